// IWR1843 range -> Doppler -> elevation -> azimuth FFT chain for gfx950 (HBM-bound).
//
// What it computes is the reference's RadarObject.generateHeatmap
// (preprocessing/process_iwr1843.py:106-173); how it computes it is not:
//
//   K1 hupr_k_range_doppler  one workgroup per (sensor-frame, virtual antenna)
//        - coalesced int16 I/Q loads of the antenna's 64 chirps (TDM demux = index math)
//        - 256-pt range FFT per chirp: LDS-staged radix-4 DIF, one wave per chirp
//        - only the 64 kept range bins (94..31) are scattered into an LDS Doppler tile
//        - clutter removal (chirp mean) + 64-pt Doppler FFT (radix-4, 16 lanes per FFT)
//        - only the 16 kept Doppler bins are written: RD[sf][12 antennas][i][r]  (98 KB / sf)
//   K2 hupr_k_angle           one workgroup per (sensor-frame, Doppler bin)
//        - the zero-padded 8x64 angle FFT has <= 12 non-zero inputs, so it is evaluated as a
//          pruned DFT:  out[e',a'] = P[a'] + w8^e' Q[a'] + [e'=0] R[a']
//        - fftshift / flip / crop of the reference collapse into the output index map
//          out[i,r,a,e] = M5[(3-e)%8, (31-a)%64, (56+i)%64, 94-r]      
//        - each lane owns one azimuth bin and stores its 8 elevation bins as one 64-B run,
//          so a wave writes 4 KB contiguous
//        - LOADER variant: keeps Doppler 4..11, splits re/im and applies the per-elevation
//          Normalize (datasets/base.py:13-24) == (x-mean)/std_unbiased of the (range,az) plane;
//          statistics come from a first pass over the same LDS-resident inputs (recompute
//          instead of a second trip through HBM)
//
// Algorithmic HBM bytes per sensor-frame: 786 432 read + 4 194 304 (c64) or 2 097 152 (loader)
// written; the RD intermediate adds 2 x 98 304.
#include "hupr_common.h"
#include "twiddles.h"

namespace hupr {

typedef float f32x4 __attribute__((ext_vector_type(4)));     // native vector: one global_store_dwordx4, never scalarised

// A wave exchanging data with ITSELF through LDS needs no workgroup barrier: DS instructions of one wave execute in
// order.  This only stops the compiler from reordering the LDS accesses across the exchange point.
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

constexpr int kRx = 4, kChirps = 192, kSamples = 256;
constexpr int kVant = 12;            // 8 azimuth + 4 elevation virtual antennas
constexpr int kRange = 64, kDop = 16, kAz = 64, kEl = 8;
constexpr int kRangeHi = 94;         // kept range bins 94,93,...,31
constexpr int kDopStride = 68;       // LDS row pitch (float2) of the Doppler tile: 8 rows x 4 lanes of a quad-per-FFT read cover all 64 banks

__device__ __forceinline__ int pad32(int p) { return p + (p >> 5); }

// radix-4 DIF butterfly, forward transform (e^{-2 pi i / N})
__device__ __forceinline__ void r4(float2& x0, float2& x1, float2& x2, float2& x3) {
    float2 a0 = cadd(x0, x2), a1 = csub(x0, x2), a2 = cadd(x1, x3), a3 = csub(x1, x3);
    float2 b3 = make_float2(a3.y, -a3.x);   // -i * a3
    x0 = cadd(a0, a2);
    x1 = cadd(a1, b3);
    x2 = csub(a0, a2);
    x3 = csub(a1, b3);
}

// 16-point forward DFT in registers: two radix-4 DIF stages.  In: x[n] natural order; out: X[g + 4 q] at x[4 g + q].
__device__ __forceinline__ void fft16(float2 (&x)[16]) {
    // W_16^m, m = 0..9 (only products j*q with j, q in 0..3 occur)
    constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
    const float2 w16[10] = {{1.f, 0.f}, {c1, -s1}, {h, -h}, {s1, -c1}, {0.f, -1.f}, {-s1, -c1}, {-h, -h}, {-c1, -s1}, {-1.f, 0.f},
                            {-c1, s1}};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r4(x[j], x[j + 4], x[j + 8], x[j + 12]);
        if (j > 0) {
            x[j + 4] = cmul(x[j + 4], w16[j]);
            x[j + 8] = cmul(x[j + 8], w16[2 * j]);
            x[j + 12] = cmul(x[j + 12], w16[3 * j]);
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) r4(x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]);
}

// ------------------------------------------------------------------------------------------
// K1: range FFT + crop + clutter removal + Doppler FFT + crop
// grid = n_sf * 12, block = 256
// ------------------------------------------------------------------------------------------
// WIN (opt-in, OFF for every parity path — the reference applies no window, process_iwr1843.py:130-134): bit 0 = Hann over
// the 256 range samples, bit 1 = Hann over the 64 chirp loops (after the clutter-removal mean), both in np.hanning's
// symmetric form 0.5 - 0.5 cos(2 pi n / (N - 1)).  The range window commutes with the chirp mean, so it is applied to the
// samples as they are converted; the Doppler window multiplies the mean-free range profile right before its FFT.
__device__ __forceinline__ float hann(int n, int N) { return 0.5f - 0.5f * cospif(2.0f * (float)n / (float)(N - 1)); }

template <int WIN>
__global__ __launch_bounds__(256) void hupr_k_range_doppler(const int16_t* __restrict__ iq,
                                                            float2* __restrict__ rd) {
    __shared__ float2 tw[256];                       // W_256^t
    __shared__ float2 rbuf_x[16 * 273];              // one 16 x 17 transpose tile per (wave, chirp slot)
    __shared__ float2 dop[kRange * kDopStride];      // [range bin][chirp loop]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sf = blockIdx.x / kVant, vant = blockIdx.x % kVant;
    // TDM demux (reference :113-120): chirp%3==0 -> az rows 0..3, ==2 -> az rows 4..7, ==1 -> elevation
    const int rx = vant & 3;
    const int tx = (vant < 4) ? 0 : (vant < 8 ? 2 : 1);

    tw[tid] = kTw256[tid];
    __syncthreads();

    // each int32 holds one (I,Q) sample
    const int32_t* src = reinterpret_cast<const int32_t*>(iq) +
                         ((size_t)(sf * kRx + rx) * kChirps) * kSamples;
    // 256-point range FFT = 16 x 16 (n = 16 n1 + n2, k = k1 + 16 k2): sixteen lanes per chirp, four chirps per wave at
    // a time.  Lane n2 holds x[16 n1 + n2] (n1 = 0..15), does the 16-point DFT over n1 in registers, applies
    // W_256^{n2 k1}, the 16 x 16 tile is transposed through LDS ONCE, and lane k1 finishes with the DFT over n2 —
    // one LDS exchange per chirp instead of three radix-4 exchanges (two of which were 2- and 4-way bank-conflicted).
    const int l16 = lane & 15, cs = lane >> 4;
    // All 16 chirps of this wave are requested up front (64 VGPRs): one HBM round trip (~2 us) is several times longer
    // than a 256-point FFT, so a one-chirp-ahead prefetch leaves every FFT waiting on its loads.
    int32_t raw[4][16];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int32_t* row = src + (size_t)(3 * (wave * 16 + ps * 4 + cs) + tx) * kSamples + l16;
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) raw[ps][n1] = row[16 * n1];
    }
    // per-lane twiddles W_256^{n2 k1}, k1 = 1..15 (independent of the chirp: read once)
    float2 twl[16];
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) twl[k1] = tw[(l16 * k1) & 255];
    float wr[(WIN & 1) ? 16 : 1];
    if constexpr (WIN & 1) {
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) wr[n1] = hann(16 * n1 + l16, kSamples);
    }
    float2* xch = rbuf_x + (wave * 4 + cs) * 273;   // 16 x 17 transpose tile of this lane's chirp slot (+1: bank offset between slots)
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int cc = wave * 16 + ps * 4 + cs;
        float2 x[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1)
            x[n1] = make_float2((float)(int16_t)(raw[ps][n1] & 0xffff), (float)(raw[ps][n1] >> 16));
        if constexpr (WIN & 1) {
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) { x[n1].x *= wr[n1]; x[n1].y *= wr[n1]; }
        }
        fft16(x);                                        // Y[k1 = g + 4 q] at x[4 g + q]
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k1 = g + 4 * q;
                xch[k1 * 17 + l16] = (k1 == 0) ? x[4 * g + q] : cmul(x[4 * g + q], twl[k1]);
            }
        wave_lds_fence();
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) x[n2] = xch[l16 * 17 + n2];     // lane k1 = l16 gathers its row
        wave_lds_fence();                                // the tile is rewritten by the next pass
        fft16(x);                                        // X[k1 + 16 k2], k2 = g + 4 q at x[4 g + q]
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = l16 + 16 * (g + 4 * q);
                const int r = kRangeHi - k;
                if (r >= 0 && r < kRange) dop[r * kDopStride + cc] = x[4 * g + q];
            }
    }
    __syncthreads();       // every wave's columns of the Doppler tile are in place

    // Doppler: 64 FFTs (one per kept range bin) of 64 points = 16 x 4 (n = 4 n1 + n2, k = k1 + 16 k2), four lanes per FFT:
    // lane n2 does the 16-point DFT over n1 in registers and applies W_64^{n2 k1}; the remaining 4-point DFT over n2 is a
    // sum across the quad (DPP), and only k2 = 0 (bins 0..7) and k2 = 3 (bins 56..63, factor i^{n2}) are kept — no LDS
    // exchange at all (the radix-4 version needed two, both bank-conflicted).
    {
        const int n2 = tid & 3, r = tid >> 2;
        const float2* row = dop + r * kDopStride;
        float2 x[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) x[n1] = row[4 * n1 + n2];
        // static clutter removal (reference :122-128): subtract the mean over the 64 chirp loops
        float sx = 0.f, sy = 0.f;
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) { sx += x[n1].x; sy += x[n1].y; }
        sx += __shfl_xor(sx, 1, 64); sy += __shfl_xor(sy, 1, 64);
        sx += __shfl_xor(sx, 2, 64); sy += __shfl_xor(sy, 2, 64);
        sx *= (1.0f / 64.0f);
        sy *= (1.0f / 64.0f);
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) { x[n1].x -= sx; x[n1].y -= sy; }
        if constexpr (WIN & 2) {
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) { const float w = hann(4 * n1 + n2, 64); x[n1].x *= w; x[n1].y *= w; }
        }
        fft16(x);                                        // Y[k1 = g + 4 q] at x[4 g + q]
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k1 = g + 4 * q;
                float2 y = x[4 * g + q];
                if (k1 > 0) y = cmul(y, tw[(4 * n2 * k1) & 255]);          // W_64^{n2 k1}
                if (k1 >= 8) {                                             // k2 = 3: W_4^{3 n2} = i^{n2}
                    const float2 t = y;
                    if (n2 == 1) y = make_float2(-t.y, t.x);
                    else if (n2 == 2) y = make_float2(-t.x, -t.y);
                    else if (n2 == 3) y = make_float2(t.y, -t.x);
                }
                y.x += __shfl_xor(y.x, 1, 64); y.y += __shfl_xor(y.y, 1, 64);
                y.x += __shfl_xor(y.x, 2, 64); y.y += __shfl_xor(y.y, 2, 64);
                // Doppler bin d = k1 (k1 < 8) or k1 + 48; fftshift + keep 24..39 -> i = (d + 8) & 63 = 0..15
                const int i = (k1 < 8) ? k1 + 8 : k1 - 8;
                // RD[sf][antenna][doppler][range]: a workgroup's 64 range bins form a 512-byte run (antenna-innermost
                // made every 8-byte store its own partial line: 4x write amplification in the counters)
                if ((k1 & 3) == n2) rd[(((size_t)sf * kVant + vant) * kDop + i) * kRange + r] = y;
            }
    }
}

// ------------------------------------------------------------------------------------------
// Packed complex arithmetic: one complex number = one 64-bit VGPR pair, every complex add / rotate-and-add is ONE
// v_pk_add_f32 and every complex multiply TWO packed instructions (op_sel picks the re/im halves, neg_lo/neg_hi the signs).
// hipcc's SLP vectoriser pairs components of DIFFERENT complex numbers and then shuffles them back with v_mov (a quarter of
// the range-first kernel's instruction stream); written out with the modifiers there is nothing to shuffle.
// ------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));

// The swizzled operand is SRC0.  Round 4 (scripts/probes/pk_victim.hip, profiles/r04_pk_victim_race.txt): a two-source VOP3P
// instruction whose SRC1 low lane reads the high half (op_sel:[0,1]: v_pk_add_f32, v_pk_mul_f32) returns changed results in 11 %
// of the threads that share the chip with hupr_k_conv_halo_bf16<64, 64> on another stream — and never alone; the same swizzle on
// SRC0 (op_sel:[1,0]), op_sel_hi forms, and the three-source v_pk_fma_f32 forms below are clean in 1.6e8 thread-evaluations each.
// Rounds 1-3 wrote these two as (a, b op_sel:[0,1]): the loader chain then differed from its launch alone in 443 of 600 launches
// beside that convolution (scripts/fft_race.py).  Addition commutes: same bits.
__device__ __forceinline__ v2f pk_add_mi(v2f a, v2f b) {          // a - i b = (a.x + b.y, a.y - b.x)
    v2f d;
    asm("v_pk_add_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ v2f pk_add_pi(v2f a, v2f b) {          // a + i b = (a.x - b.y, a.y + b.x)
    v2f d;
    asm("v_pk_add_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// a * w:  t = (a.x w.x, a.x w.y);  d = (t.x - a.y w.y, t.y + a.y w.x).   _s: w uniform (SGPR pair), _v: w per lane.
__device__ __forceinline__ v2f pk_cmul_s(v2f a, v2f w) {
    v2f t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "s"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(a), "s"(w), "v"(t));
    return d;
}
__device__ __forceinline__ v2f pk_cmul_v(v2f a, v2f w) {
    v2f t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(a), "v"(w), "v"(t));
    return d;
}
__device__ __forceinline__ v2f pk_cfma_s(v2f acc, v2f a, v2f w) {  // acc + a * w
    v2f t, d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(t) : "v"(a), "s"(w), "v"(acc));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(a), "s"(w), "v"(t));
    return d;
}

__device__ __forceinline__ void r4p(v2f& x0, v2f& x1, v2f& x2, v2f& x3) {      // radix-4 DIF butterfly, forward
    const v2f a0 = x0 + x2, a1 = x0 - x2, a2 = x1 + x3, a3 = x1 - x3;
    x0 = a0 + a2;
    x1 = pk_add_mi(a1, a3);
    x2 = a0 - a2;
    x3 = pk_add_pi(a1, a3);
}

// 16-point forward DFT, same factorisation / output order as fft16(): X[g + 4 q] at x[4 g + q].  64 + 18 packed instructions.
__device__ __forceinline__ void fft16p(v2f (&x)[16]) {
    constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
    constexpr float wre[10] = {1.f, c1, h, s1, 0.f, -s1, -h, -c1, -1.f, -c1};
    constexpr float wim[10] = {0.f, -s1, -h, -c1, -1.f, -c1, -h, -s1, 0.f, s1};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r4p(x[j], x[j + 4], x[j + 8], x[j + 12]);
        if (j > 0) {
            x[j + 4] = pk_cmul_s(x[j + 4], (v2f){wre[j], wim[j]});
            x[j + 8] = pk_cmul_s(x[j + 8], (v2f){wre[2 * j], wim[2 * j]});
            x[j + 12] = pk_cmul_s(x[j + 12], (v2f){wre[3 * j], wim[3 * j]});
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) r4p(x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]);
}

// ------------------------------------------------------------------------------------------
// K1 (round 3): Doppler FIRST, then range — same 2-D DFT, a quarter of the range-FFT work.
// grid = n_sf * 12, block = 256 (thread = ADC sample index)
// ------------------------------------------------------------------------------------------
// The range-first kernel above runs 64 full 256-point FFTs per (sensor-frame, antenna) and throws three quarters of their
// output away; it issues ~3 100 instructions per wave and is bound by instruction issue (80 % busy), not by HBM (2.6 TB/s).
// The 2-D transform is separable, so the pruned axis goes first: the chirp-loop (Doppler) DFT keeps 16 of its 64 bins, and
// only those 16 sequences go through the 256-point range FFT.
//   stage A  thread s owns ADC sample s of all 64 chirp loops (every load instruction is one contiguous 256-byte run; raw
//            buffer loads with the chirp offset in an SGPR: no address arithmetic on the vector pipe).  64-point DFT =
//            16 x 4 (n = 4 n1 + n2) entirely in registers: for each n2 a 16-point DFT over n1, then the 16 kept bins
//            d = 0..7, 56..63 accumulate  acc[d] += Y_n2[d mod 16] * W_64^{n2 d}  (compile-time twiddles in SGPRs; nothing
//            crosses lanes).  -> LDS tile[doppler i][sample]
//            Clutter removal (reference :122-128) costs nothing and is EXACT: the samples are integers, so every sum /
//            difference of the first butterflies is exact in fp32, the chirp mean only ever reaches the DC outputs
//            Y_n2[0] (again exact sums), and subtracting it makes bin d = 0 exactly zero while every other bin is
//            bit-identical with or without it — so the unwindowed kernel just writes zero to bin 0.  (With a window the
//            products are no longer integers and the mean is subtracted explicitly.)
//   stage B  16 range FFTs per workgroup (one 16-lane group each): 256 = 16 x 16 with ONE transpose through the group's own
//            tile row, as in the range-first kernel; bins 94..31 go straight to RD[sf][antenna][i][r].
// 37 KB of LDS.  The zero-Doppler bin (i = 8) would therefore be EXACTLY zero where the range-first order (and the
// reference's fft2 of the mean-free cube, process_iwr1843.py:122-134) leaves rounding noise that the reference's Normalize
// inflates to a unit-variance channel (and an exact zero to 0/0 = NaN): by default the bin carries the dither defined
// below instead (HUPR_FFT_ZERO_DOPPLER_EXACT keeps the exact zero; the loader epilogues map a zero-variance plane to zeros).
// HALF: only the eight Doppler bins the loader keeps (i = 4..11, dataset.py:145) — half the accumulators of stage A and half
// the range FFTs (two of the four waves retire after stage A); the other rows of RD are left untouched.
// Zero-Doppler dither (round 4).  The reference's zero-Doppler bin is the fp64 rounding residue of fft2 over the mean-free
// chirps (process_iwr1843.py:122-134: ~2e-16 of the other bins, white over range and antenna, a pure function of the frame)
// and its Normalize (datasets/base.py:17-24) turns that plane into a unit-variance input channel — an exactly-zero plane
// would be 0/0 there.  No other implementation can reproduce pocketfft's residue, so this chain carries a stand-in with the
// same statistics: per (antenna, ADC sample) a pair of 16-bit integers hashed from the exact chirp sums T (the very
// quantity clutter removal cancels) and the position, scaled by 2^-53 of an ADC LSB; the range FFT of stage B turns it
// into a white, near-Gaussian row of rms ~5e-11 (the reference: 4e-11 on full-scale inputs) that the angle kernel and the
// loader epilogues treat like every other Doppler bin.  oracle/fft_chain.py::zero_doppler_dither restates it bit for bit.
__device__ __forceinline__ v2f zero_doppler_dither(v2f chirp_sum, int vant, int s) {
    uint32_t h = (uint32_t)(int)chirp_sum.x * 0x9E3779B1u ^ (uint32_t)(int)chirp_sum.y * 0x85EBCA77u ^
                 (uint32_t)(vant * 256 + s) * 0xC2B2AE3Du;
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    constexpr float kScale = 1.1102230246251565e-16f;           // 2^-53
    return (v2f){(float)(int16_t)(h & 0xffffu) * kScale, (float)((int32_t)h >> 16) * kScale};
}

template <int WIN, bool HALF, int AUX = 0>
__global__ __launch_bounds__(256) void hupr_k_doppler_range(const int16_t* __restrict__ iq, float2* __restrict__ rd, int zd_exact, int n_items,
                                                            int grouped) {
    constexpr int kPitch = 272;                      // v2f per Doppler row: 2 x 272 = 32 (mod 64) banks, and = 16 x 17
    constexpr int kRows = HALF ? 8 : kDop, kRow0 = HALF ? 4 : 0;
    __shared__ v2f tw[256];                          // W_256^t
    __shared__ v2f tile[kRows * kPitch];             // [doppler i - kRow0][sample]; a row doubles as its group's transpose scratch

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // work item = (sensor-frame, virtual antenna).  grouped: the three antennas that share a receiver (TX1 / TX3 / TX2 = chirps
    // 3 c + 0 / 2 / 1 of the SAME 192 KB block of the cube, interleaved row by row) run back to back on ONE XCD (workgroup id ->
    // XCD is id % 8), so that the three passes over a block reach the memory controller together instead of from three XCDs at
    // three different times
    int item = blockIdx.x;
    if (grouped) {
        const int x = blockIdx.x & 7, k = blockIdx.x >> 3;       // XCD, position in the XCD's sequence
        const int g = (k / 3) * 8 + x, t3 = k % 3;               // (sensor-frame, receiver) group, which of its three antennas
        item = (g >> 2) * kVant + (g & 3) + 4 * t3;
        if (item >= n_items) return;
    }
    const int sf = item / kVant, vant = item % kVant;
    const int rx = vant & 3;
    const int tx = (vant < 4) ? 0 : (vant < 8 ? 2 : 1);          // TDM demux (reference :113-120)
    tw[tid] = reinterpret_cast<const v2f*>(kTw256)[tid];

    // ---- stage A ------------------------------------------------------------------------------------------------
    constexpr int kRowBytes = kSamples * 4;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int16_t*>(iq) + (size_t)(sf * kRx + rx) * kChirps * kSamples * 2, 0, kChirps * kRowBytes, 0x00020000);
    int32_t raw[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) raw[c] = __builtin_amdgcn_raw_buffer_load_b32(rs, tid * 4, (3 * c + tx) * kRowBytes, AUX);
    v2f mean = (v2f){0.f, 0.f};
    float wr = 1.0f;
    if constexpr (WIN != 0) {
#pragma unroll
        for (int c = 0; c < 64; ++c) mean += (v2f){(float)(int16_t)(raw[c] & 0xffff), (float)(raw[c] >> 16)};
        mean *= (1.0f / 64.0f);
        if constexpr (WIN & 1) wr = hann(tid, kSamples);
    }
    v2f acc[16];
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) {
        v2f x[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            const int32_t v = raw[4 * n1 + n2];
            x[n1] = (v2f){(float)(int16_t)(v & 0xffff), (float)(v >> 16)};
            if constexpr (WIN != 0) {
                x[n1] -= mean;
                float w = wr;
                if constexpr (WIN & 2) w *= hann(4 * n1 + n2, 64);
                x[n1] *= w;
            }
        }
        fft16p(x);                                       // Y[k1 = g + 4 q] at x[4 g + q]
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k1 = g + 4 * q;
                const int d = (k1 < 8) ? k1 : k1 + 48;   // kept Doppler bin with d = k1 (mod 16)
                if (HALF && k1 >= 4 && k1 < 12) continue;
                if (n2 == 0) acc[k1] = x[4 * g + q];
                else acc[k1] = pk_cfma_s(acc[k1], x[4 * g + q], (v2f){w64re(n2 * d), w64im(n2 * d)});
            }
    }
    if constexpr (WIN == 0)                              // exact clutter removal, see above: acc[0] holds the exact chirp sum
        acc[0] = zd_exact ? (v2f){0.f, 0.f} : zero_doppler_dither(acc[0], vant, tid);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) {
        const int i = (k1 < 8) ? k1 + 8 : k1 - 8;        // fftshift + keep 24..39 -> i = (d + 8) & 63
        if (i >= kRow0 && i < kRow0 + kRows) tile[(i - kRow0) * kPitch + tid] = acc[k1];
    }
    __syncthreads();
    if (HALF && wave >= 2) return;                       // eight rows = two waves of range FFTs

    // ---- stage B: 256-point range FFT of Doppler row i, sixteen lanes per FFT ---------------------------------------
    const int l16 = lane & 15, i = kRow0 + wave * 4 + (lane >> 4);
    v2f* row = tile + (i - kRow0) * kPitch;
    v2f x[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) x[n1] = row[16 * n1 + l16];
    v2f twl[16];
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) twl[k1] = tw[(l16 * k1) & 255];
    wave_lds_fence();                                    // the row is this group's scratch from here on
    fft16p(x);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k1 = g + 4 * q;
            row[k1 * 17 + l16] = (k1 == 0) ? x[4 * g + q] : pk_cmul_v(x[4 * g + q], twl[k1]);
        }
    wave_lds_fence();
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) x[n2] = row[l16 * 17 + n2];
    fft16p(x);                                           // X[k1 + 16 k2], k1 = l16, k2 = g + 4 q at x[4 g + q]
    v2f* dst = reinterpret_cast<v2f*>(rd) + (((size_t)sf * kVant + vant) * kDop + i) * kRange;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k2 = g + 4 * q;
            if (k2 < 1 || k2 > 5) continue;              // bins 94..31 live in k2 = 1..5 only
            const int r = kRangeHi - (l16 + 16 * k2);
            if (r >= 0 && r < kRange) dst[r] = x[4 * g + q];
        }
}

// ------------------------------------------------------------------------------------------
// K2: pruned angle DFT + index map (+ optional loader epilogue)
// grid = n_sf * (LOADER ? 8 : 16), block = 256 (4 waves x 16 range cells each)
// ------------------------------------------------------------------------------------------
// The angle kernel exists in four instantiations (complex cube, loader, magnitude, loader + elevation mean) that must
// produce the SAME bits for the same cell: its arithmetic is written as explicit packed instructions (pk_* helpers above), so
// there is nothing for the compiler to contract or re-associate differently per instantiation.
#pragma clang fp contract(off)

__device__ __forceinline__ v2f pk_cfma_v(v2f acc, v2f a, v2f w) {  // acc + a * w, w per lane
    v2f t, d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(t) : "v"(a), "v"(w), "v"(acc));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(a), "v"(w), "v"(t));
    return d;
}

struct AngleOut {
    v2f o[kEl];   // indexed by OUTPUT elevation bin e
};

// One (range, azimuth-bin) cell: the zero-padded 8 x 64 angle FFT with <= 12 non-zero inputs, as a pruned DFT
//   out[e', a'] = P[a'] + w8^e' Q[a'] + [e' = 0] R[a'],   output elevation bin e <- e' = (3 - e) mod 8.
// w8^e' Q takes only Q, -iQ and U = h (1 - i) Q: the eight outputs are eight rotate-and-add instructions.
__device__ __forceinline__ AngleOut angle_cell(const v2f* __restrict__ cell, const v2f* twl) {
    // cell[0..7] azimuth antennas, cell[8..11] elevated antennas (at azimuth offsets 2..5); twl[a] = W_64^{a a'}
    v2f P = pk_cmul_v(cell[2], twl[2]), Q = pk_cmul_v(cell[8], twl[2]);
#pragma unroll
    for (int a = 3; a < 6; ++a) {
        P = pk_cfma_v(P, cell[a], twl[a]);
        Q = pk_cfma_v(Q, cell[8 + a - 2], twl[a]);
    }
    v2f R = pk_cfma_v(cell[0], cell[1], twl[1]);
    R = pk_cfma_v(R, cell[6], twl[6]);
    R = pk_cfma_v(R, cell[7], twl[7]);
    constexpr float h = 0.70710678118654752440f;
    const v2f U = pk_add_mi(Q, Q) * h;          // w8^1 Q = h (1 - i) Q
    AngleOut r;
    r.o[0] = pk_add_mi(P, U);                   // e' = 3: -i U
    r.o[1] = pk_add_mi(P, Q);                   // e' = 2: -i Q
    r.o[2] = P + U;                             // e' = 1
    r.o[3] = (P + Q) + R;                       // e' = 0
    r.o[4] = pk_add_pi(P, U);                   // e' = 7: +i U
    r.o[5] = pk_add_pi(P, Q);                   // e' = 6: +i Q
    r.o[6] = P - U;                             // e' = 5
    r.o[7] = P - Q;                             // e' = 4
    return r;
}

// MODE 0: complex64 cube (the reference's output); 1: loader epilogue (Normalize fused); 2 (opt-in): magnitude |X| as fp32;
// 3: loader epilogue + HuPRNet's elevation mean (models/networks.py:26-27) — writes the (re/im, Doppler) planes
// means[sf][2 f + c][range][azimuth] that the MNet front end consumes: 1/8 of the loader's bytes, and MNet no longer re-reads them
//
// Normalize statistics in closed form (round 3; the first version evaluated the whole pruned DFT twice).  A (re/im, elevation)
// plane is, per range bin r, the 64-point DFT over a' of  z_r[a] = p_r[a] + w8^e' q_r[a] + [e' = 0] rest_r[a]  (a = 0..7), so
//   sum_a' X        = 64 z_r[0]                                   (only e' = 0 has z[0] != 0)
//   sum_a' X^2      = 64 sum_a z[a] z[-a]     = 64 z_r[0]^2       (a = 1..7 pair with the zero inputs 63..57)
//   sum_a' |X|^2    = 64 sum_a |z[a]|^2                           (Parseval)
//   sum_a' Re(X)^2  = (sum |X|^2 + Re sum X^2) / 2,  sum_a' Im(X)^2 = (sum |X|^2 - Re sum X^2) / 2
// and  sum_a |p + w q|^2 = sum (|p|^2 + |q|^2) + 2 Re(w sum conj(p) q):  five real sums over the 64 x 12 inputs of the
// workgroup give the mean and the unbiased variance of all sixteen planes.
// 8 workgroups per CU (<= 64 registers per lane): the loader grids — 8 planes x 256 sensor-frames = 2 048 workgroups — are then ONE
// round of the 256 CUs instead of 1.14 (70 registers: 7 per CU, a second round of 256 workgroups at one per CU)
template <int MODE>
__global__ __launch_bounds__(256) void hupr_k_angle(const float2* __restrict__ rd, void* __restrict__ out_) {
    constexpr bool LOADER = MODE == 1 || MODE == 3;
    __shared__ v2f cells[kRange * kVant];      // RD for this (sf, i): 64 range bins x 12 antennas
    __shared__ v2f tw64[64];
    __shared__ v2f s_mean[8], s_rstd[8];       // (re, im) pairs per OUTPUT elevation bin
    __shared__ double s_red[8];                // the seven sums behind the statistics, one or two per wave

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int kPlanes = LOADER ? 8 : 16;
    const int sf = blockIdx.x / kPlanes, pl = blockIdx.x % kPlanes;
    const int i = LOADER ? pl + 4 : pl;         // loader keeps Doppler indices 4..11 (dataset.py:145)

    // RD[sf][antenna][doppler][range] -> cells[range][antenna]: twelve 512-byte runs
    for (int t = tid; t < kRange * kVant; t += 256) {
        const int v = t >> 6, r = t & 63;
        cells[r * kVant + v] = reinterpret_cast<const v2f*>(rd)[(((size_t)sf * kVant + v) * kDop + i) * kRange + r];
    }
    if (tid < 64) tw64[tid] = reinterpret_cast<const v2f*>(kTw256)[4 * tid];
    __syncthreads();

    if (LOADER) {
        {                                       // lane = range bin; wave w reduces its share of the seven sums (same per-lane
            const v2f* c = cells + lane * kVant;    // expressions and the same butterfly order as when one wave did all seven)
            double ra = 0.0, rb = 0.0;
            if (wave == 0) {                    // sa
#pragma unroll
                for (int a = 2; a < 6; ++a) {
                    const double px = c[a].x, py = c[a].y, qx = c[8 + a - 2].x, qy = c[8 + a - 2].y;
                    ra += px * px + py * py + qx * qx + qy * qy;
                }
            } else if (wave == 1) {             // conj(p) q
#pragma unroll
                for (int a = 2; a < 6; ++a) {
                    const double px = c[a].x, py = c[a].y, qx = c[8 + a - 2].x, qy = c[8 + a - 2].y;
                    ra += px * qx + py * qy;
                    rb += px * qy - py * qx;
                }
            } else if (wave == 2) {             // the e' = 0 extras
                const double t4 = (double)c[1].x * c[1].x + (double)c[1].y * c[1].y + (double)c[6].x * c[6].x + (double)c[6].y * c[6].y +
                                  (double)c[7].x * c[7].x + (double)c[7].y * c[7].y;
                const double z0x = c[0].x, z0y = c[0].y;
                ra = t4 + z0x * z0x + z0y * z0y;
                rb = z0x * z0x - z0y * z0y;
            } else {
                ra = c[0].x;
                rb = c[0].y;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) ra += __shfl_xor(ra, o, 64);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) rb += __shfl_xor(rb, o, 64);
            if (lane == 0) {
                s_red[wave == 0 ? 0 : wave == 1 ? 1 : wave == 2 ? 3 : 4] = ra;      // red[0..6] = sa, scx, scy, t4 + |z0|^2, z0x, z0y, Re z0^2
                s_red[wave == 0 ? 7 : wave == 1 ? 2 : wave == 2 ? 6 : 5] = rb;
            }
        }
        __syncthreads();
        if (wave == 0) {
            double red[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) red[k] = s_red[k];
            if (lane < 8) {                     // lane = OUTPUT elevation bin e <- e' = (3 - e) mod 8
                const int ep = (3 - lane) & 7;
                constexpr double hh = 0.70710678118654752440;
                const double w8x[8] = {1.0, hh, 0.0, -hh, -1.0, -hh, 0.0, hh}, w8y[8] = {0.0, -hh, -1.0, -hh, 0.0, hh, 1.0, hh};
                double wx = 0.0, wy = 0.0;
#pragma unroll
                for (int k = 0; k < 8; ++k) if (k == ep) { wx = w8x[k]; wy = w8y[k]; }
                double e2 = red[0] + 2.0 * (wx * red[1] - wy * red[2]);        // sum_r sum_a |z|^2
                double sx = 0.0, sy = 0.0, zz = 0.0;
                if (ep == 0) { e2 += red[3]; sx = 64.0 * red[4]; sy = 64.0 * red[5]; zz = red[6]; }
                const double n = 4096.0;
                const double ssx = 32.0 * (e2 + zz), ssy = 32.0 * (e2 - zz);
                const double mx = sx / n, my = sy / n;
                const double vx = (ssx - sx * mx) / (n - 1.0), vy = (ssy - sy * my) / (n - 1.0);   // unbiased, like torch.std_mean
                s_mean[lane] = (v2f){(float)mx, (float)my};
                // an exactly-zero plane (the zero-Doppler bin of the Doppler-first chain) normalises to zeros, not 0/0
                s_rstd[lane] = (v2f){vx > 0.0 ? (float)(1.0 / sqrt(vx)) : 0.0f, vy > 0.0 ? (float)(1.0 / sqrt(vy)) : 0.0f};
            }
        }
        __syncthreads();
    }

    // lane == azimuth-FFT output bin a'; twl[a] = W_64^{a a'}
    v2f twl[8];
#pragma unroll
    for (int a = 1; a < 8; ++a) twl[a] = tw64[(a * lane) & 63];
    const int a_out = (31 - lane) & 63;

    if (LOADER) {
        // x * rstd - mean * rstd, one fused multiply-add per (re, im) pair.  The sixteen 1/std are the same in every lane: held in
        // scalar registers (the multiply-add takes one scalar operand) they leave the kernel at <= 64 vector registers = 8
        // workgroups per CU, so that the 2 048 workgroups of a 256-sensor-frame loader call are ONE round of the chip, not 1.14
        v2f rstd[8], nmr[8];
#pragma unroll
        for (int e = 0; e < kEl; ++e) {
            // (written as asm: through __builtin_amdgcn_readfirstlane hipcc read only the .x components and broadcast them into
            // both halves of the packed operand — op_sel_hi:[1,0] on an s[n:n+1] whose upper half was never written;
            // tests/test_fft_gpu.py::test_fused_loader_vs_oracle caught it)
            const v2f r = s_rstd[e];
            float rx, ry;
            asm volatile("v_readfirstlane_b32 %0, %2\n\tv_readfirstlane_b32 %1, %3" : "=s"(rx), "=s"(ry) : "v"(r.x), "v"(r.y));
            rstd[e] = (v2f){rx, ry};
            nmr[e] = -(s_mean[e] * rstd[e]);
        }
        float* out = reinterpret_cast<float*>(out_);
        if constexpr (MODE == 3) {
            // means[sf][j = 2 pl + c][r][a] = mean over the 8 elevation bins of the normalised plane, summed exactly like
            // hupr_k_mnet_fwd sums the stored loader tensor (same association -> the two paths agree bit for bit)
            float* mre = out + ((size_t)(sf * 16 + 2 * pl) * kRange) * kAz;
            float* mim = mre + (size_t)kRange * kAz;
            for (int j = 0; j < 16; ++j) {
                const int r = wave * 16 + j;
                AngleOut v = angle_cell(cells + r * kVant, twl);
                v2f n[8];
#pragma unroll
                for (int e = 0; e < kEl; ++e) n[e] = __builtin_elementwise_fma(v.o[e], rstd[e], nmr[e]);
                const v2f m = (((n[0] + n[1]) + (n[2] + n[3])) + ((n[4] + n[5]) + (n[6] + n[7]))) * 0.125f;
                mre[r * kAz + a_out] = m.x;
                mim[r * kAz + a_out] = m.y;
            }
            return;
        }
        // out[sf][f=pl][c][r][a][e]
        float* base_re = out + ((size_t)((sf * 8 + pl) * 2 + 0) * kRange) * kAz * kEl;
        float* base_im = base_re + (size_t)kRange * kAz * kEl;
        for (int j = 0; j < 16; ++j) {
            const int r = wave * 16 + j;
            AngleOut v = angle_cell(cells + r * kVant, twl);
            v2f n[8];
#pragma unroll
            for (int e = 0; e < kEl; ++e) n[e] = __builtin_elementwise_fma(v.o[e], rstd[e], nmr[e]);
            f32x4* pr = reinterpret_cast<f32x4*>(base_re + ((size_t)r * kAz + a_out) * kEl);
            f32x4* pi = reinterpret_cast<f32x4*>(base_im + ((size_t)r * kAz + a_out) * kEl);
            pr[0] = (f32x4){n[0].x, n[1].x, n[2].x, n[3].x};
            pr[1] = (f32x4){n[4].x, n[5].x, n[6].x, n[7].x};
            pi[0] = (f32x4){n[0].y, n[1].y, n[2].y, n[3].y};
            pi[1] = (f32x4){n[4].y, n[5].y, n[6].y, n[7].y};
        }
    } else if (MODE == 2) {
        float* out = reinterpret_cast<float*>(out_) + ((size_t)(sf * kDop + i) * kRange) * kAz * kEl;
        for (int j = 0; j < 16; ++j) {
            const int r = wave * 16 + j;
            AngleOut v = angle_cell(cells + r * kVant, twl);
            float m[kEl];
#pragma unroll
            for (int e = 0; e < kEl; ++e) m[e] = sqrtf(fmaf(v.o[e].x, v.o[e].x, v.o[e].y * v.o[e].y));
            f32x4* p = reinterpret_cast<f32x4*>(out + ((size_t)r * kAz + a_out) * kEl);
            p[0] = (f32x4){m[0], m[1], m[2], m[3]};
            p[1] = (f32x4){m[4], m[5], m[6], m[7]};
        }
    } else {
        float2* out = reinterpret_cast<float2*>(out_) + ((size_t)(sf * kDop + i) * kRange) * kAz * kEl;
        for (int j = 0; j < 16; ++j) {
            const int r = wave * 16 + j;
            AngleOut v = angle_cell(cells + r * kVant, twl);
            f32x4* p = reinterpret_cast<f32x4*>(out + ((size_t)r * kAz + a_out) * kEl);
#pragma unroll
            for (int e = 0; e < kEl; e += 2)
                p[e >> 1] = (f32x4){v.o[e].x, v.o[e].y, v.o[e + 1].x, v.o[e + 1].y};
        }
    }
}

// ------------------------------------------------------------------------------------------
// (a2) loader glue on a precomputed complex cube: Doppler select + re/im split + Normalize
// grid = n_sf * 8, block = 256; each block owns the 16 planes of one (sf, f)
// cube[sf][16][64][64][8] float2 -> out[sf][8][2][64][64][8] float
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hupr_k_loader_normalize(const float2* __restrict__ cube,
                                                               float* __restrict__ out) {
    __shared__ float red[4][32];
    __shared__ float s_mean[16], s_rstd[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sf = blockIdx.x >> 3, f = blockIdx.x & 7, i = f + 4;
    // view the (64,64,8) complex slab as float4 = 2 complex (elevation pair)
    const float4* src = reinterpret_cast<const float4*>(cube + ((size_t)(sf * kDop + i)) * kRange * kAz * kEl);
    constexpr int kVec = kRange * kAz * kEl / 2;     // 16384 float4
    const int epair = tid & 3;                        // which elevation pair this thread always sees
    float sum[4] = {0, 0, 0, 0}, ssq[4] = {0, 0, 0, 0};   // (e0.re, e0.im, e1.re, e1.im)
    for (int t = tid; t < kVec; t += 256) {
        float4 v = src[t];
        sum[0] += v.x; ssq[0] = fmaf(v.x, v.x, ssq[0]);
        sum[1] += v.y; ssq[1] = fmaf(v.y, v.y, ssq[1]);
        sum[2] += v.z; ssq[2] = fmaf(v.z, v.z, ssq[2]);
        sum[3] += v.w; ssq[3] = fmaf(v.w, v.w, ssq[3]);
    }
    // reduce across lanes that share the same elevation pair (lane & 3): xor 4,8,16,32
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int o = 32; o >= 4; o >>= 1) {
            sum[k] += __shfl_xor(sum[k], o, 64);
            ssq[k] += __shfl_xor(ssq[k], o, 64);
        }
    }
    if (lane < 4) {
        // plane index k16 = c*8 + e, with e = 2*epair + {0,1}
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = 2 * epair + (k >> 1), c = k & 1;
            red[wave][c * 8 + e] = sum[k];
            red[wave][16 + c * 8 + e] = ssq[k];
        }
    }
    __syncthreads();
    if (tid < 16) {
        const double n = 4096.0;
        double S = (double)red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        double SS = (double)red[0][16 + tid] + red[1][16 + tid] + red[2][16 + tid] + red[3][16 + tid];
        double mean = S / n;
        double var = (SS - S * mean) / (n - 1.0);
        s_mean[tid] = (float)mean;
        s_rstd[tid] = var > 0.0 ? (float)(1.0 / sqrt(var)) : 0.0f;      // exactly-zero plane (zero Doppler): zeros, not 0/0
    }
    __syncthreads();
    const int e0 = 2 * epair;
    const float m0r = s_mean[e0], m1r = s_mean[e0 + 1], m0i = s_mean[8 + e0], m1i = s_mean[8 + e0 + 1];
    const float r0r = s_rstd[e0], r1r = s_rstd[e0 + 1], r0i = s_rstd[8 + e0], r1i = s_rstd[8 + e0 + 1];
    float* ore = out + ((size_t)((sf * 8 + f) * 2 + 0)) * kRange * kAz * kEl;
    float* oim = ore + (size_t)kRange * kAz * kEl;
    for (int t = tid; t < kVec; t += 256) {
        float4 v = src[t];                       // second read is L2-resident (256 KB slab)
        float2 re = make_float2((v.x - m0r) * r0r, (v.z - m1r) * r1r);
        float2 im = make_float2((v.y - m0i) * r0i, (v.w - m1i) * r1i);
        reinterpret_cast<float2*>(ore)[t] = re;
        reinterpret_cast<float2*>(oim)[t] = im;
    }
}

}  // namespace hupr

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
using namespace hupr;

extern "C" size_t hupr_fft_chain_ws_bytes(int n_sf) {
    if (n_sf <= 0) return 0;
    return (size_t)n_sf * kDop * kRange * kVant * sizeof(float2);
}

// A/B aid.  bit 0 = TEMPORAL ADC loads (rounds 1-3), default non-temporal: the cube is read exactly once, and with the nt hint the
// cold pass — how the training step sees it, 25 GB of other traffic since the last call — runs as fast as the Infinity-Cache-warm
// one: 106.4 -> 68.6 us per 256 sensor-frames, 2.52 -> 3.91 TB/s on SURVEY's bytes (profiles/r04_fft_variants.txt).
// bit 1 = the three antennas of a receiver back to back on one XCD (measured neutral, off).

static int fft_chain_common(const int16_t* adc_iq, int n_sf, void* out, void* ws, size_t ws_bytes,
                            hupr_stream_t stream, bool loader, int flags = 0, bool means = false) {
    HUPR_REQUIRE((flags & ~(HUPR_FFT_HANN_RANGE | HUPR_FFT_HANN_DOPPLER | HUPR_FFT_MAGNITUDE | HUPR_FFT_ZERO_DOPPLER_EXACT |
                            HUPR_FFT_RANGE_FIRST)) == 0, "hupr_fft_chain: flags=0x%x", flags);
    HUPR_REQUIRE(!((flags & HUPR_FFT_ZERO_DOPPLER_EXACT) && (flags & HUPR_FFT_RANGE_FIRST)),
                 "hupr_fft_chain: the range-first order has no exact zero-Doppler form");
    HUPR_REQUIRE(!(loader && (flags & HUPR_FFT_MAGNITUDE)), "hupr_fft_chain: the loader epilogue splits re/im, it has no magnitude form");
    HUPR_REQUIRE(n_sf >= 0, "hupr_fft_chain: n_sf=%d", n_sf);
    if (n_sf == 0) return HUPR_OK;                      // empty batch is a no-op
    HUPR_REQUIRE(adc_iq && out && ws, "hupr_fft_chain: null pointer");
    HUPR_REQUIRE(((uintptr_t)adc_iq & 3) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)ws & 15) == 0,
                 "hupr_fft_chain: misaligned buffer");
    HUPR_REQUIRE(n_sf <= (1 << 20), "hupr_fft_chain: n_sf=%d too large", n_sf);
    if (ws_bytes < hupr_fft_chain_ws_bytes(n_sf))
        return fail(HUPR_ERR_WORKSPACE, "hupr_fft_chain: workspace %zu < %zu", ws_bytes,
                    hupr_fft_chain_ws_bytes(n_sf));
    hipStream_t s = as_stream(stream);
    float2* rd = reinterpret_cast<float2*>(ws);
    const int zd = (flags & HUPR_FFT_ZERO_DOPPLER_EXACT) ? 1 : 0;
    if (flags & HUPR_FFT_RANGE_FIRST) {
        switch (flags & 3) {
            case 0: HUPR_LAUNCH(hupr_k_range_doppler<0>, dim3(n_sf * kVant), dim3(256), 0, s, adc_iq, rd); break;
            case 1: HUPR_LAUNCH(hupr_k_range_doppler<1>, dim3(n_sf * kVant), dim3(256), 0, s, adc_iq, rd); break;
            case 2: HUPR_LAUNCH(hupr_k_range_doppler<2>, dim3(n_sf * kVant), dim3(256), 0, s, adc_iq, rd); break;
            default: HUPR_LAUNCH(hupr_k_range_doppler<3>, dim3(n_sf * kVant), dim3(256), 0, s, adc_iq, rd); break;
        }
    } else {
        const int n_items = n_sf * kVant;
        const dim3 g1(n_items), b1(256);
#define HUPR_DR(W_, H_) HUPR_LAUNCH((hupr_k_doppler_range<W_, H_, 2>), g1, b1, 0, s, adc_iq, rd, zd, n_items, 0)
        switch ((flags & 3) | (loader ? 4 : 0)) {
            case 0: HUPR_DR(0, false); break;
            case 1: HUPR_DR(1, false); break;
            case 2: HUPR_DR(2, false); break;
            case 3: HUPR_DR(3, false); break;
            case 4: HUPR_DR(0, true); break;
            case 5: HUPR_DR(1, true); break;
            case 6: HUPR_DR(2, true); break;
            default: HUPR_DR(3, true); break;
        }
#undef HUPR_DR
    }
    HUPR_LAUNCH_OK("hupr_k_range_doppler");
    if (loader && means)
        HUPR_LAUNCH(hupr_k_angle<3>, dim3(n_sf * 8), dim3(256), 0, s, rd, out);
    else if (loader)
        HUPR_LAUNCH(hupr_k_angle<1>, dim3(n_sf * 8), dim3(256), 0, s, rd, out);
    else if (flags & HUPR_FFT_MAGNITUDE)
        HUPR_LAUNCH(hupr_k_angle<2>, dim3(n_sf * 16), dim3(256), 0, s, rd, out);
    else
        HUPR_LAUNCH(hupr_k_angle<0>, dim3(n_sf * 16), dim3(256), 0, s, rd, out);
    HUPR_LAUNCH_OK("hupr_k_angle");
    return HUPR_OK;
}

extern "C" int hupr_fft_chain_c64(const int16_t* adc_iq, int n_sf, void* out_c64, void* ws,
                                  size_t ws_bytes, hupr_stream_t stream) {
    return fft_chain_common(adc_iq, n_sf, out_c64, ws, ws_bytes, stream, false);
}

extern "C" int hupr_fft_chain_loader_f32(const int16_t* adc_iq, int n_sf, float* out, void* ws,
                                         size_t ws_bytes, hupr_stream_t stream) {
    return fft_chain_common(adc_iq, n_sf, out, ws, ws_bytes, stream, true);
}

extern "C" int hupr_fft_chain_loader_means_f32(const int16_t* adc_iq, int n_sf, float* means, void* ws, size_t ws_bytes,
                                               hupr_stream_t stream) {
    return fft_chain_common(adc_iq, n_sf, means, ws, ws_bytes, stream, true, 0, true);
}

extern "C" int hupr_fft_chain_opts(const int16_t* adc_iq, int n_sf, void* out, int flags, int loader, void* ws, size_t ws_bytes,
                                   hupr_stream_t stream) {
    HUPR_REQUIRE(loader >= 0 && loader <= 2, "hupr_fft_chain_opts: loader=%d", loader);
    return fft_chain_common(adc_iq, n_sf, out, ws, ws_bytes, stream, loader != 0, flags, loader == 2);
}

extern "C" int hupr_loader_normalize_c64(const void* cube_c64, int n_sf, float* out, hupr_stream_t stream) {
    HUPR_REQUIRE(n_sf >= 0, "hupr_loader_normalize_c64: n_sf=%d", n_sf);
    if (n_sf == 0) return HUPR_OK;
    HUPR_REQUIRE(cube_c64 && out, "hupr_loader_normalize_c64: null pointer");
    HUPR_REQUIRE(((uintptr_t)cube_c64 & 15) == 0 && ((uintptr_t)out & 15) == 0,
                 "hupr_loader_normalize_c64: misaligned buffer");
    HUPR_LAUNCH(hupr_k_loader_normalize, dim3(n_sf * 8), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float2*>(cube_c64), out);
    HUPR_LAUNCH_OK("hupr_k_loader_normalize");
    return HUPR_OK;
}

// ------------------------------------------------------------------------------------------
// DCA1000 raw capture -> device ADC layout (reference getadcDataFromDCA1000, process_iwr1843.py:54-83)
// raw: int16 groups [I(2k), I(2k+1), Q(2k), Q(2k+1)]; per chirp the complex stream is [rx0 x256][rx1 x256][rx2 x256][rx3 x256].
// out: int16 [frame][rx][chirp 192][sample 256][I,Q].  One thread moves one 8-byte group (two samples): coalesced
// 8-byte reads, 8-byte writes.
// ------------------------------------------------------------------------------------------
namespace hupr {
__global__ __launch_bounds__(256) void hupr_k_dca1000_deinterleave(const short4* __restrict__ raw, short4* __restrict__ out,
                                                                   long n_groups) {
    for (long gidx = (long)blockIdx.x * 256 + threadIdx.x; gidx < n_groups; gidx += (long)gridDim.x * 256) {
        const short4 v = raw[gidx];                       // I0 I1 Q0 Q1 of stream samples n = 2g, 2g+1
        const long n = gidx * 2;
        const int s = (int)(n & 255);                     // sample index inside the rx run (even)
        const long run = n >> 8;                          // = chirp * 4 + rx
        const int rx = (int)(run & 3);
        const long chirp = run >> 2;
        const long frame = chirp / kChirps;
        const int c = (int)(chirp - frame * kChirps);
        const long dst = (((frame * kRx + rx) * kChirps + c) * kSamples + s) * 2;      // int16 index, multiple of 4
        out[dst >> 2] = make_short4(v.x, v.z, v.y, v.w);  // (I0,Q0,I1,Q1)
    }
}
}  // namespace hupr

extern "C" int hupr_dca1000_deinterleave(const int16_t* raw, int16_t* adc_iq, int n_frames, hupr_stream_t stream) {
    HUPR_REQUIRE(n_frames >= 0, "hupr_dca1000_deinterleave: n_frames=%d", n_frames);
    if (n_frames == 0) return HUPR_OK;
    HUPR_REQUIRE(raw && adc_iq && ((uintptr_t)raw & 7) == 0 && ((uintptr_t)adc_iq & 7) == 0,
                 "hupr_dca1000_deinterleave: null or misaligned pointer");
    const long n_groups = (long)n_frames * kRx * kChirps * kSamples / 2;
    HUPR_LAUNCH(hupr::hupr_k_dca1000_deinterleave, dim3((unsigned)min((long)8192, (n_groups + 255) / 256)), dim3(256), 0,
                       as_stream(stream), reinterpret_cast<const short4*>(raw), reinterpret_cast<short4*>(adc_iq), n_groups);
    HUPR_LAUNCH_OK("hupr_k_dca1000_deinterleave");
    return HUPR_OK;
}
