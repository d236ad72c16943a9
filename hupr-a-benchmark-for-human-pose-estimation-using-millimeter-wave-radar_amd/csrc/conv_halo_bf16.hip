// LDS halo-tiled 3x3x3 / 1x3x3 "same" convolution on the bf16 matrix pipe (fp32 tensors, fp32 accumulate).
//
// The implicit-GEMM kernel (gemm_bf16.hip) re-gathers every im2col row for each of the 27 taps, so at
// bf16 MFMA rates it is bound by global->LDS bytes (~23 B/clk/CU sustained, ~12 % matrix utilisation).
// Here a workgroup owns a spatial tile of 128 output voxels (2x8x8, or 1x8x16 for 2-D maps) and a
// 64- (or 32-) wide slice of output channels:
//   * the input halo ((TD+kd-1) x 10 x (TW+2) voxels x KC channels) is loaded ONCE per channel
//     chunk, rounded to bf16 and kept in LDS; all taps read their shifted A fragments from it
//     (lane = output voxel, address = halo base + tap offset, ds_read_b128 of 8 channels);
//   * the per-tap weight tile [BN][KC] (pre-packed bf16, L2 resident) is double-buffered through LDS,
//     next tap's tile in flight during the MFMAs, one barrier per tap;
//   * A bytes from global drop from 27x to ~3x the tile, so the kernel is matrix/LDS bound.
// Used for forward and (with tap-reversed, transposed weights) the input gradient.
#include "conv_halo.h"

namespace hupr {



// Tile geometry (compile time): 3-D layers 2 x 8 x 8 voxels with a 4 x 10 x 10 halo and 27 taps; 2-D maps 1 x 8 x 16 with a
// 1 x 10 x 18 halo and 9 taps.  A stage = the three ky taps of one (kz, kx) column, so that a lane whose two output rows
// are neighbours in y reads the four halo rows hy .. hy+3 once for all six (row, ky) products (see conv_halo256m_bf16.hip).
template <int BN, int KC, bool ABF, bool IS3D>
__global__ __launch_bounds__(256, 2) void hupr_k_conv_halo_bf16(HaloArgs p) {      // two workgroups per CU (LDS <= 75 KB each)
    // KC = 64: unpadded 128-byte rows whose 16-byte chunks are XOR-swizzled — halo rows by
    //   key = ((hx >> 1) & 3) | (((hy >> 1) & 1) << 2), weight rows by key = (n >> 1) & 7 — which makes every
    //   ds_read_b128 lane group hit all 64 banks for every tap shift.  KC = 32 (only the Cin = 32 stem): padded 80-byte rows.
    constexpr bool SWZ = (KC == 64);
    constexpr int LDK = SWZ ? KC : KC + 8;            // bf16 elements per LDS row
    constexpr int WN = (BN == 64) ? 2 : 1, WM = 4 / WN;
    constexpr int TM = (128 / WM) / 32;               // 32-voxel accumulator tiles per wave: 2 (BN = 64) or 1 (BN = 32)
    constexpr int C8 = KC / 8;                        // 8-channel groups per row
    constexpr int TS = 3;
    constexpr int B_LD = (TS * BN * C8 + 255) / 256;  // 16-byte weight loads per thread per stage
    constexpr int TD = IS3D ? 2 : 1, TW = IS3D ? 8 : 16, KD = IS3D ? 3 : 1;
    constexpr int HD = TD + KD - 1, HH = 10, HW = TW + 2;
    constexpr int NVOX = HD * HH * HW;                // 400 (3-D) / 180 (2-D)
    constexpr int T = KD * 9;
    constexpr int NI = (NVOX * C8 + 255) / 256;       // halo items (8 channels of one voxel) per thread

    __shared__ __attribute__((aligned(16))) __bf16 Hs[NVOX * LDK];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[TS][BN * LDK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;

    // block -> (spatial tile, co tile); co fastest so consecutive workgroups reuse the same halo via L2
    // XCD-aware order: workgroup b runs on XCD b % 8, so each XCD gets a contiguous run of tiles and spatial
    // neighbours (which share halo voxels) meet in the same L2 instead of re-fetching through the fabric.
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int cot = bid % p.n_co_tiles;
    int st = bid / p.n_co_tiles;
    const int twi = st % p.nw; st /= p.nw;
    const int thi = st % p.nh; st /= p.nh;
    const int tdi = st % p.nd;
    const int b = st / p.nd;
    const int d0 = tdi * TD, h0 = thi * 8, w0 = twi * TW;
    const int n0 = cot * BN;

    // MFMA column lr of accumulator tile i <-> output voxel (dz, hy0 + i, wx):
    //   TM = 2: 3-D  dz = wm,      hy0 = 2 (lr >> 3),            wx = lr & 7      2-D  hy0 = 4 wm + 2 (lr >> 4), wx = lr & 15
    //   TM = 1: 3-D  dz = wm >> 1, hy0 = 4 (wm & 1) + (lr >> 3), wx = lr & 7      2-D  hy0 = 2 wm + (lr >> 4),   wx = lr & 15
    const int wx = IS3D ? (lr & 7) : (lr & 15);
    const int eb = IS3D ? (lr >> 3) : (lr >> 4);
    const int dz = IS3D ? (TM == 2 ? wm : (wm >> 1)) : 0;
    const int hy0 = (TM == 2) ? ((IS3D ? 0 : 4 * wm) + 2 * eb) : (IS3D ? 4 * (wm & 1) + eb : 2 * wm + eb);
    const int abase = ((dz * HH + hy0) * HW + wx) * LDK;       // halo element of tap (0,0,0) of accumulator tile 0
    const int bkey = ((wn * 32 + lr) >> 1) & 7;               // weight row swizzle key of this lane's B fragment row

    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // weight-stage loads of this thread: item f = tid + 256 j over [ky t][row n][chunk c8]; rows past Co are clamped
    // (their output columns are never stored) so that the loads stay branch-free and hipcc keeps counted vmcnt waits
    const __bf16* bsrc[B_LD];
    int bdst[B_LD];
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
        const int f = (tid + 256 * j) % (TS * BN * C8);
        const int t = f / (BN * C8), r = f % (BN * C8), n = r / C8, c8 = r % C8;
        bsrc[j] = p.wp + (long)min(n0 + n, p.Co - 1) * T * p.Ci + (long)t * 3 * p.Ci + c8 * 8;     // tap = (kz*3 + ky)*3 + kx
        bdst[j] = t * (BN * LDK) + n * LDK + (SWZ ? (((c8 ^ (n >> 1)) & 7) << 3) : c8 * 8);
    }
    u32x4 rb[B_LD];                                   // next stage's weight tiles, in flight during the MFMAs
    __bf16* const Bflat = &Bs[0][0];

    // K slice of this workgroup (p.part != nullptr): units = (channel chunk, kz plane), unit u = chunk * KD + kz
    const int n_units = (p.Ci / KC) * KD;
    const int u_begin = p.part ? (int)blockIdx.y * p.units_per_slice : 0;
    const int u_end = p.part ? min(n_units, u_begin + p.units_per_slice) : n_units;
    bool first_chunk = true;
    for (int c0 = 0; c0 < p.Ci; c0 += KC) {
        const int cu = (c0 / KC) * KD;
        const int kz_lo = max(0, u_begin - cu), kz_hi = min(KD, u_end - cu);
        if (kz_lo >= kz_hi) continue;           // this chunk belongs to other slices (workgroup-uniform)
#pragma unroll
        for (int j = 0; j < B_LD; ++j) rb[j] = *reinterpret_cast<const u32x4*>(bsrc[j] + (long)(kz_lo * 9) * p.Ci + c0);      // first stage: kz = kz_lo, kx = 0
        if (!first_chunk) __syncthreads();      // previous chunk's readers are done with Hs / Bs
        first_chunk = false;
        // ---- halo chunk: global -> bf16 LDS, zero outside the tensor.  ALL of this thread's loads are issued before any
        // is stored, so the fill costs about one memory round trip. ---------------------------------------------------
        if (!(p.ablate & 1)) {
            constexpr int NB_ = IS3D ? (NI + 1) / 2 : NI;      // 3-D halo (13 items): two batches keep the kernel at 2 workgroups / CU
#pragma unroll
            for (int ub = 0; ub < NI; ub += NB_) {
                u32x4 vb[ABF ? NB_ : 1];
                float4 va[ABF ? 1 : NB_], vc[ABF ? 1 : NB_];
#pragma unroll
                for (int u = 0; u < NB_; ++u) {
                    const int it = tid + (ub + u) * 256;
                    if constexpr (ABF) vb[u] = (u32x4){0u, 0u, 0u, 0u};
                    else { va[u] = make_float4(0.f, 0.f, 0.f, 0.f); vc[u] = va[u]; }
                    if (it < NVOX * C8) {
                        const int vox = it / C8, c8 = it - vox * C8;
                        const int hx = vox % HW;
                        const int t = vox / HW;
                        const int hy = t % HH, hz = t / HH;
                        const int d = d0 + hz - (KD >> 1), h = h0 + hy - 1, w = w0 + hx - 1;
                        if ((unsigned)d < (unsigned)p.D && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W) {
                            const long off = ((((long)b * p.D + d) * p.H + h) * p.W + w) * p.in_ld + c0 + c8 * 8;
                            if constexpr (ABF) {
                                vb[u] = *reinterpret_cast<const u32x4*>(static_cast<const __bf16*>(p.x) + off);
                            } else {
                                const float* src = static_cast<const float*>(p.x) + off;
                                va[u] = *reinterpret_cast<const float4*>(src);
                                vc[u] = *reinterpret_cast<const float4*>(src + 4);
                            }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < NB_; ++u) {
                    const int it = tid + (ub + u) * 256;
                    if (it < NVOX * C8) {
                        const int vox = it / C8, c8 = it - vox * C8;
                        const int hx = vox % HW, hy = (vox / HW) % HH;
                        __bf16* dstp = &Hs[vox * LDK + (SWZ ? (c8 ^ (((hx >> 1) & 3) | (((hy >> 1) & 1) << 2))) : c8) * 8];
                        if constexpr (ABF) {
                            *reinterpret_cast<u32x4*>(dstp) = vb[u];
                        } else {
                            bf16x8 v;
                            v[0] = (__bf16)va[u].x; v[1] = (__bf16)va[u].y; v[2] = (__bf16)va[u].z; v[3] = (__bf16)va[u].w;
                            v[4] = (__bf16)vc[u].x; v[5] = (__bf16)vc[u].y; v[6] = (__bf16)vc[u].z; v[7] = (__bf16)vc[u].w;
                            *reinterpret_cast<bf16x8*>(dstp) = v;
                        }
                    }
                }
            }
        }
        // ---- stages: compute from Bs while the next stage's weights travel to registers -------------------------------
        // (kz is a real loop — unrolling all nine 3-D stages costs ~60 VGPRs and spills at two workgroups per CU)
#pragma unroll 1
        for (int kz = kz_lo; kz < kz_hi; ++kz)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int st_ = kz * 3 + kx;
#pragma unroll
            for (int j = 0; j < B_LD; ++j)
                if (B_LD * 256 == TS * BN * C8 || tid + 256 * j < TS * BN * C8) *reinterpret_cast<u32x4*>(&Bflat[bdst[j]]) = rb[j];
            __syncthreads();
            if (st_ + 1 < kz_hi * 3) {
                const long soff = (long)(((st_ + 1) / 3) * 9 + ((st_ + 1) % 3)) * p.Ci + c0;     // tap (kz, ky = 0, kx) of the next stage
#pragma unroll
                for (int j = 0; j < B_LD; ++j) rb[j] = *reinterpret_cast<const u32x4*>(bsrc[j] + soff);
            }
            {
                const int toff = (kz * HH * HW + kx) * LDK;
                const int xkey = ((wx + kx) >> 1) & 3;
                constexpr int NA = TM + 2;                     // halo rows hy0 .. hy0 + TM + 1 of this lane's column
                bf16x8 af[2][NA], bq[2][TS];                   // two fragment sets: K-step ks + 1 is read while ks multiplies
#define HUPR_FRAGS(SET_, KS_)                                                                                       \
                {                                                                                                   \
                    const int cw_ = (KS_) * 2 + lh;                                                                 \
                    _Pragma("unroll") for (int r = 0; r < NA; ++r)                                                  \
                        af[SET_][r] = *reinterpret_cast<const bf16x8*>(                                             \
                            &Hs[abase + toff + r * (HW * LDK) +                                                     \
                                (SWZ ? (cw_ ^ (xkey | ((((hy0 + r) >> 1) & 1) << 2))) : cw_) * 8]);                  \
                    _Pragma("unroll") for (int t = 0; t < TS; ++t)                                                  \
                        bq[SET_][t] = *reinterpret_cast<const bf16x8*>(                                             \
                            &Bs[t][(wn * 32 + lr) * LDK + (SWZ ? (cw_ ^ bkey) : cw_) * 8]);                          \
                }
                HUPR_FRAGS(0, 0)
#pragma unroll
                for (int ks = 0; ks < KC / 16; ++ks) {
                    if (ks + 1 < KC / 16) {
                        if (ks & 1) { HUPR_FRAGS(0, ks + 1) } else { HUPR_FRAGS(1, ks + 1) }
                    }
                    // (as in conv_halo256m_bf16.hip: the reads of K-step ks + 1 are spread in front of the MFMAs of ks instead of
                    // issued as one burst — NA + TS reads against TS * TM MFMAs)
#pragma unroll
                    for (int t = 0; t < TS; ++t)               // ky;  D'[channel][voxel]
#pragma unroll
                        for (int i = 0; i < TM; ++i)
                            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[ks & 1][t], af[ks & 1][t + i], acc[i], 0, 0, 0);
#pragma unroll
                    for (int i_ = 0; i_ < TS * TM; ++i_) {
                        __builtin_amdgcn_sched_group_barrier(0x100, TM == 2 ? 1 : 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#undef HUPR_FRAGS
            }
            __syncthreads();                      // all waves are done with Bs (and, on the last stage, with Hs)
        }
    }

    // ---- epilogue (conv_halo.h): this lane's voxel of each 32-voxel group, 4 x 4 channels ----
    if (!(p.ablate & 4)) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const long m = (((long)b * p.D + d0 + dz) * p.H + h0 + hy0 + i) * p.W + w0 + wx;
            if (p.part) halo_store_partial(p, acc[i], m, n0 + wn * 32 + 4 * lh, blockIdx.y);      // K slice: finished by the reduce kernel
            else halo_store_voxel<ABF>(p, acc[i], m, n0 + wn * 32 + 4 * lh);
        }
    }
}

// y[m][c] = bf16( sum_s part[s][m][c] (+ bias[c]) (+ res[m][c]) ) in slice order: the second half of a K-sliced convolution.
__global__ __launch_bounds__(256) void hupr_k_conv_partial_reduce(HaloArgs p, int n_slices) {
    const long M = (long)p.Bn * p.D * p.H * p.W;
    const int c4n = p.Co >> 2;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * c4n) return;
    const long m = i / c4n;
    const int ch = (int)(i - m * c4n) * 4;
    f32x4n v = *reinterpret_cast<const f32x4n*>(p.part + m * p.Co + ch);
    for (int s = 1; s < n_slices; ++s) v += *reinterpret_cast<const f32x4n*>(p.part + ((long)s * M + m) * p.Co + ch);
    if (p.bias) v += (f32x4n){p.bias[ch], p.bias[ch + 1], p.bias[ch + 2], p.bias[ch + 3]};
    if (p.res) {
        const bf16x4 rv = *reinterpret_cast<const bf16x4*>(static_cast<const __bf16*>(p.res) + m * p.res_ld + ch);
        v += (f32x4n){(float)rv[0], (float)rv[1], (float)rv[2], (float)rv[3]};
    }
    const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    *reinterpret_cast<bf16x4*>(static_cast<__bf16*>(p.y) + m * p.out_ld + ch) = o;
}

// w (Co, Ci, taps) fp32 parameter layout -> bf16  mode 0: [Co][tap][Ci]   mode 1: [Ci][taps-1-tap][Co]
__global__ void hupr_k_pack_weights_bf16(const float* __restrict__ w, __bf16* __restrict__ wp, int co_n, int ci_n,
                                         int taps, int mode) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)co_n * ci_n * taps;
    if (i >= n) return;
    if (mode == 0) {
        const int ci = i % ci_n;
        const long t = i / ci_n;
        const int tap = t % taps, co = t / taps;
        wp[i] = (__bf16)w[((long)co * ci_n + ci) * taps + tap];
    } else {
        const int co = i % co_n;
        const long t = i / co_n;
        const int tapf = t % taps, ci = t / taps;
        wp[i] = (__bf16)w[((long)co * ci_n + ci) * taps + (taps - 1 - tapf)];
    }
}

// Table-driven repack of MANY convolution weights in one launch (both layouts of every entry): a training step otherwise
// issues ~160 six-microsecond pack launches.  One thread per source element; the entry is found by binary search.
struct PackDesc {          // mirrors the packed struct built in functional.py (little-endian, 48 bytes)
    const float* w;        // (Co, Ci, taps) parameter
    void* wp0;             // [Co][tap][Ci]
    void* wp1;             // [Ci][taps-1-tap][Co]
    long first;            // index of this entry's first element in the concatenated element space
    int co, ci, taps, kind;   // kind 0: fp32 outputs, 1: bf16 outputs; 2 / 3: projection-concatenation entries (block layout 3)
};
// Block kinds (blocks[b] = {entry, layout, start}):
//   layout 0 / 1: 2048 consecutive DESTINATION elements of that layout of the entry, starting at `start` — the generic
//                 element-wise form (coalesced writes, 4-byte gathers): small or odd-shaped weights, fp32 outputs;
//   layout 2    : one 32 (co) x 32 (ci) x taps tile of a bf16-output entry with Co % 32 == 0, Ci % 32 == 0, taps <= 27,
//                 start = co0 << 32 | ci0.  The tile is read as 32 contiguous runs of 32*taps floats, parked in LDS as
//                 bf16, and leaves as 64-byte runs of BOTH layouts (16 bytes per lane) — the element-wise form spent
//                 its time on 4-byte gathers strided by Ci*taps for the transposed layout.
struct PackBlock { int entry, layout; long start; };
constexpr int kPackTile = 32, kPackMaxTaps = 27;
__global__ __launch_bounds__(256) void hupr_k_pack_table(const PackDesc* __restrict__ descs, const PackBlock* __restrict__ blocks) {
    __shared__ __bf16 tile[kPackTile * (kPackTile * kPackMaxTaps + 2)];
    const PackBlock pb = blocks[blockIdx.x];
    const PackDesc d = descs[pb.entry];
    const int tid = threadIdx.x;
    if (pb.layout == 2) {
        const int co0 = (int)(pb.start >> 32), ci0 = (int)(pb.start & 0xffffffff);
        const int T = d.taps, run = kPackTile * T, pitch = run + 2;        // [co][ci][tap] with a padded row
        if ((reinterpret_cast<uintptr_t>(d.w) & 15) == 0) {
            // 16-byte loads, four per thread in flight (a tile row is 32 T contiguous floats, 128-byte aligned inside an aligned tensor):
            // with two 55 KB blocks per CU the 4-byte form had 2 KB per wave in flight and ran the 284 MB of a step's repack at 2 TB/s
            typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
            const int run4 = run >> 2, n4 = kPackTile * run4;
            for (int i0 = tid; i0 < n4; i0 += 1024) {
                f32x4n v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = i0 + 256 * u;
                    if (idx < n4) {
                        const int r = idx / run4, o = (idx - r * run4) << 2;
                        v[u] = *reinterpret_cast<const f32x4n*>(d.w + ((long)(co0 + r) * d.ci + ci0) * T + o);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = i0 + 256 * u;
                    if (idx < n4) {
                        const int r = idx / run4, o = (idx - r * run4) << 2;
                        bf16x2_* dst = reinterpret_cast<bf16x2_*>(&tile[r * pitch + o]);      // pitch and o are even: 4-byte aligned
                        dst[0] = (bf16x2_){(__bf16)v[u][0], (__bf16)v[u][1]};
                        dst[1] = (bf16x2_){(__bf16)v[u][2], (__bf16)v[u][3]};
                    }
                }
            }
        } else {
            for (int idx = tid; idx < kPackTile * run; idx += 256) {
                const int r = idx / run, o = idx - r * run;
                tile[r * pitch + o] = (__bf16)d.w[((long)(co0 + r) * d.ci + ci0) * T + o];
            }
        }
        __syncthreads();
        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
        __bf16* wp0 = static_cast<__bf16*>(d.wp0);
        __bf16* wp1 = static_cast<__bf16*>(d.wp1);
        const int v8 = (tid & 3) * 8;                                      // 8 consecutive ci (layout 0) / co (layout 1)
        for (int rt = tid >> 2; rt < kPackTile * T; rt += 64) {
            const int r = rt / T, t = rt - r * T;                          // layout 0: row (co0 + r, tap t), ci0 + v8 ..
            bf16x8 v;
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = tile[r * pitch + (v8 + k) * T + t];
            *reinterpret_cast<bf16x8*>(wp0 + ((long)(co0 + r) * T + t) * d.ci + ci0 + v8) = v;
            const int i = r, tf = t;                                       // layout 1: row (ci0 + i, flipped tap tf), co0 + v8 ..
            bf16x8 u;
#pragma unroll
            for (int k = 0; k < 8; ++k) u[k] = tile[(v8 + k) * pitch + i * T + (T - 1 - tf)];
            *reinterpret_cast<bf16x8*>(wp1 + ((long)(ci0 + i) * T + tf) * d.co + co0 + v8) = u;
        }
        return;
    }
    const long count = (long)d.co * d.ci * d.taps;
    if (pb.layout == 3) {
        // a 1x1 projection weight of an MSCSA level (models/layers.py:150-157) copied into its row block of the level's two
        // concatenated (4C, C) fp32 matrices: wp0 = the plain one (backward GEMMs), wp1 = the one the forward projection reads,
        // whose query rows (kind 3) carry the factor log2(e) of the QS attention kernels; kind 2: both plain
        const float f = d.kind == 3 ? 1.4426950408889634f : 1.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long l = pb.start + tid + 256 * u;
            if (l >= count) break;
            const float v = d.w[l];
            static_cast<float*>(d.wp0)[l] = v;
            static_cast<float*>(d.wp1)[l] = v * f;
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const long l = pb.start + tid + 256 * u;
        if (l >= count) break;
        long src;
        if (pb.layout == 0) {                           // [co][tap][ci]
            const int ci = (int)(l % d.ci);
            const long t = l / d.ci;
            const int tap = (int)(t % d.taps), co = (int)(t / d.taps);
            src = ((long)co * d.ci + ci) * d.taps + tap;
        } else {                                        // [ci][taps-1-tap][co]
            const int co = (int)(l % d.co);
            const long t = l / d.co;
            const int tapf = (int)(t % d.taps), ci = (int)(t / d.taps);
            src = ((long)co * d.ci + ci) * d.taps + (d.taps - 1 - tapf);
        }
        const float v = d.w[src];
        void* dst = pb.layout ? d.wp1 : d.wp0;
        if (d.kind == 1) static_cast<__bf16*>(dst)[l] = (__bf16)v;
        else static_cast<float*>(dst)[l] = v;
    }
}

}  // namespace hupr

using namespace hupr;

extern "C" int hupr_pack_conv_weights_table(const void* descs_dev, const void* blocks_dev, int n_blocks, hupr_stream_t stream) {
    HUPR_REQUIRE(descs_dev && blocks_dev && n_blocks > 0, "hupr_pack_conv_weights_table: bad argument");
    static_assert(sizeof(PackDesc) == 48 && sizeof(PackBlock) == 16, "PackDesc / PackBlock layouts are part of the ABI");
    HUPR_LAUNCH(hupr_k_pack_table, dim3((unsigned)n_blocks), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const PackDesc*>(descs_dev), reinterpret_cast<const PackBlock*>(blocks_dev));
    HUPR_LAUNCH_OK("hupr_k_pack_table");
    return HUPR_OK;
}

extern "C" int hupr_pack_conv_weights_bf16(const float* w, void* wp_bf16, int Co, int Ci, int taps, int mode,
                                           hupr_stream_t stream) {
    HUPR_REQUIRE(w && wp_bf16 && Co > 0 && Ci > 0 && taps > 0 && (mode == 0 || mode == 1),
                 "hupr_pack_conv_weights_bf16: bad argument");
    const long n = (long)Co * Ci * taps;
    HUPR_LAUNCH(hupr_k_pack_weights_bf16, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), w,
                       reinterpret_cast<__bf16*>(wp_bf16), Co, Ci, taps, mode);
    HUPR_LAUNCH_OK("hupr_k_pack_weights_bf16");
    return HUPR_OK;
}

static int g_halo_ablate = 0;
static int g_halo_variant = 0;      // 0 auto, 1 force the 128-voxel kernel (A/B comparisons)
static unsigned long long* g_halo_trace = nullptr;
extern "C" void hupr_debug_halo_trace(void* buf) { g_halo_trace = reinterpret_cast<unsigned long long*>(buf); }
extern "C" void hupr_debug_halo_ablate(int bits) { g_halo_ablate = bits; }   // profiling aids (scripts/halo_ablation.py)
extern "C" void hupr_debug_halo_variant(int v) { g_halo_variant = v; }
extern "C" void hupr_debug_halo_tiles(int mask) { set_halo_tiles(mask); }      // 256-voxel kernel: bit 0 / 1 / 2 = its 4 x 8 x 8 / 2 x 8 x 16 / 1 x 16 x 16 tile in use
static int g_halo_split_k = 1;      // A/B aid: 0 = never slice the reduction of small grids
extern "C" void hupr_debug_halo_split_k(int on) { g_halo_split_k = on; }

// 1 if hupr_conv3x3_halo_bf16 supports this geometry (else use hupr_conv_fwd_bf16)
extern "C" int hupr_conv3x3_halo_supported(int D, int H, int W, int Ci, int kd, int kh, int kw, int pd, int ph, int pw) {
    if (kh != 3 || kw != 3 || ph != 1 || pw != 1) return 0;
    if (!((kd == 3 && pd == 1) || (kd == 1 && pd == 0))) return 0;
    if (Ci % 32 != 0 || H % 8 != 0) return 0;
    if (kd == 3) return (D % 2 == 0 && W % 8 == 0) ? 1 : 0;
    return (D == 1 && W % 16 == 0) ? 1 : 0;
}

// K slices of the 128-voxel kernel for grids that leave most of the chip idle (single-sample inference: a level-3 layer is
// 32 workgroups walking 36 stages each).  Returns the slice count (1 = no split) and the units per slice.
static int halo_k_slices(int Bn, int D, int H, int W, int Ci, int Co, int kd, int* units_per_slice) {
    *units_per_slice = 0;
    if (!g_halo_split_k || !hupr_conv3x3_halo_supported(D, H, W, Ci, kd, 3, 3, kd / 2, 1, 1) || Co % 32 != 0) return 1;
    const int td = kd == 3 ? 2 : 1, tw = kd == 3 ? 8 : 16;
    const long tiles = (long)Bn * (D / td) * (H / 8) * (W / tw);
    if (tiles * ((Co + 63) / 64) > 256) return 1;                 // enough workgroups (and possibly the persistent kernels)
    const long blocks = tiles * (Co / 32);                        // small grids run 32-wide channel tiles
    const int units = (Ci / (Ci % 64 == 0 ? 64 : 32)) * (kd == 3 ? 3 : 1);
    if (blocks >= 384 || units < 2) return 1;
    int s = (int)min((long)units, (768 + blocks - 1) / blocks);
    const int per = (units + s - 1) / s;
    s = (units + per - 1) / per;
    *units_per_slice = per;
    return s;
}

static int g_halo_no_res_prefetch = 0;
extern "C" void hupr_debug_halo_res_prefetch(int on) { g_halo_no_res_prefetch = on ? 0 : 1; }

static int conv3x3_halo(const void* x, const void* wp_bf16, const float* bias, const void* res, void* y, int Bn, int D,
                        int H, int W, int Ci, int in_ld, int Co, int out_ld, int res_ld, int kd, bool abf,
                        hupr_stream_t stream, const char* who, double* stats = nullptr, void* ws = nullptr, size_t ws_bytes = 0,
                        bool partial_only = false) {
    HUPR_REQUIRE(x && wp_bf16 && y, "%s: null pointer", who);
    HUPR_REQUIRE(hupr_conv3x3_halo_supported(D, H, W, Ci, kd, 3, 3, kd / 2, 1, 1), "%s: unsupported geometry", who);
    const int al = abf ? 8 : 4;                      // 16-byte halo loads, 4-channel output vectors
    HUPR_REQUIRE(Bn > 0 && Co > 0 && Co % 4 == 0 && in_ld % al == 0 && out_ld % 4 == 0 && (!res || res_ld % 4 == 0),
                 "%s: bad argument (Co, leading dimensions must be multiples of 4; bf16 in_ld of 8)", who);
    HaloArgs a;
    a.x = x; a.wp = reinterpret_cast<const __bf16*>(wp_bf16); a.bias = bias; a.res = res; a.y = y;
    a.Bn = Bn; a.D = D; a.H = H; a.W = W; a.Ci = Ci; a.in_ld = in_ld; a.Co = Co; a.out_ld = out_ld; a.res_ld = res_ld;
    a.kd = kd;
    a.ablate = g_halo_ablate;
    a.trace = g_halo_trace;
    a.stats = stats;
    a.part = nullptr;
    a.units_per_slice = 0;
    a.no_res_prefetch = g_halo_no_res_prefetch;
    if (stats) {
        HUPR_REQUIRE(abf && !bias && !res && conv_halo256_stats_ok(a, Bn),
                     "%s: fused statistics need the 256-voxel kernel (see hupr_conv3x3_halo_stats_supported), no bias / residual", who);
        HUPR_REQUIRE(launch_conv_halo256(a, Bn, abf, as_stream(stream)), "%s: 256-voxel kernel refused the launch", who);
        HUPR_LAUNCH_OK("hupr_k_conv_halo256m_bf16<stats>");
        return HUPR_OK;
    }
    // variants: 0 auto (the 256-voxel kernel where one of its forms applies, else the 128-voxel kernel), 1 force the 128-voxel kernel (A/B comparisons)
    if (!partial_only && g_halo_variant != 1 && launch_conv_halo256(a, Bn, abf, as_stream(stream))) {
        HUPR_LAUNCH_OK("hupr_k_conv_halo256m_bf16");
        return HUPR_OK;
    }
    if (kd == 3) { a.TD = 2; a.log2TW = 3; } else { a.TD = 1; a.log2TW = 4; }
    a.nd = D / a.TD; a.nh = H / 8; a.nw = W >> a.log2TW;
    // 32-wide channel tiles for Co <= 32 — and for grids that would leave CUs idle or single-tiled with 64-wide ones (small
    // batches, e.g. the B = 1 inference): twice the workgroups, two of them per CU (LDS), each with half the weight traffic
    const long blocks64 = (long)Bn * a.nd * a.nh * a.nw * ((Co + 63) / 64);
    const bool n32 = (Co <= 32) || (blocks64 <= 256 && Co % 32 == 0);
    const int bn = n32 ? 32 : 64;
    a.n_co_tiles = (Co + bn - 1) / bn;
    const long blocks = (long)Bn * a.nd * a.nh * a.nw * a.n_co_tiles;
    HUPR_REQUIRE(blocks < (1L << 31), "%s: grid too large", who);
    hipStream_t s = as_stream(stream);
    int n_slices = 1;
    if (ws && abf && n32) {                                       // K-sliced form (bf16 activations, caller supplied the workspace)
        int per = 0;
        const int want = halo_k_slices(Bn, D, H, W, Ci, Co, kd, &per);
        const size_t need = (size_t)want * Bn * D * H * W * Co * sizeof(float);
        if (want > 1 && ws_bytes >= need && ((uintptr_t)ws & 15) == 0) {
            n_slices = want;
            a.part = static_cast<float*>(ws);
            a.units_per_slice = per;
        }
    }
    HUPR_REQUIRE(!partial_only || n_slices > 1,
                 "%s: this geometry is not K-sliced (hupr_conv3x3_halo_splitk_ws_bytes() == 0) or the workspace is too small", who);
#define HUPR_HALO_LAUNCH(BN_, KC_)                                                                                       \
    do {                                                                                                                 \
        const dim3 grid_((unsigned)blocks, (unsigned)n_slices);                                                          \
        if (kd == 3) {                                                                                                   \
            if (abf) HUPR_LAUNCH((hupr_k_conv_halo_bf16<BN_, KC_, true, true>), grid_, dim3(256), 0, s, a);       \
            else HUPR_LAUNCH((hupr_k_conv_halo_bf16<BN_, KC_, false, true>), grid_, dim3(256), 0, s, a);          \
        } else {                                                                                                         \
            if (abf) HUPR_LAUNCH((hupr_k_conv_halo_bf16<BN_, KC_, true, false>), grid_, dim3(256), 0, s, a);      \
            else HUPR_LAUNCH((hupr_k_conv_halo_bf16<BN_, KC_, false, false>), grid_, dim3(256), 0, s, a);         \
        }                                                                                                                \
    } while (0)
    if (Ci % 64 == 0) {
        if (n32) HUPR_HALO_LAUNCH(32, 64); else HUPR_HALO_LAUNCH(64, 64);
    } else {
        if (n32) HUPR_HALO_LAUNCH(32, 32); else HUPR_HALO_LAUNCH(64, 32);
    }
#undef HUPR_HALO_LAUNCH
    HUPR_LAUNCH_OK("hupr_k_conv_halo_bf16");
    if (partial_only) return HUPR_OK;
    if (n_slices > 1) {
        const long n4 = (long)Bn * D * H * W * (Co / 4);
        HUPR_LAUNCH(hupr_k_conv_partial_reduce, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, a, n_slices);
        HUPR_LAUNCH_OK("hupr_k_conv_partial_reduce");
    }
    return HUPR_OK;
}

// y = conv3x3(x) (+bias) (+res): stride 1, "same" padding; kd = 3 (pad 1) or kd = 1.
// wp_bf16: weights packed by hupr_pack_conv_weights_bf16 ([Co][kd*9][Ci]).  fp32 activations in HBM.
extern "C" int hupr_conv3x3_halo_bf16(const float* x, const void* wp_bf16, const float* bias, const float* res, float* y,
                                      int Bn, int D, int H, int W, int Ci, int in_ld, int Co, int out_ld, int res_ld,
                                      int kd, hupr_stream_t stream) {
    return conv3x3_halo(x, wp_bf16, bias, res, y, Bn, D, H, W, Ci, in_ld, Co, out_ld, res_ld, kd, false, stream,
                        "hupr_conv3x3_halo_bf16");
}

// Same operator on bf16 activations (x, res, y stored as bf16; leading dimensions in elements).
extern "C" int hupr_conv3x3_halo_bf16act(const void* x, const void* wp_bf16, const float* bias, const void* res, void* y,
                                         int Bn, int D, int H, int W, int Ci, int in_ld, int Co, int out_ld, int res_ld,
                                         int kd, hupr_stream_t stream) {
    return conv3x3_halo(x, wp_bf16, bias, res, y, Bn, D, H, W, Ci, in_ld, Co, out_ld, res_ld, kd, true, stream,
                        "hupr_conv3x3_halo_bf16act");
}

// The same operator with a caller-supplied workspace: small grids (single-sample inference) split the reduction over
// (channel chunk, kz plane) slices that leave fp32 partial sums in `ws`, summed in slice order (+ bias, + residual, one rounding)
// by a second launch.  hupr_conv3x3_halo_splitk_ws_bytes() is 0 where the one-launch form is used anyway (then ws may be null).
extern "C" size_t hupr_conv3x3_halo_splitk_ws_bytes(int Bn, int D, int H, int W, int Ci, int Co, int kd) {
    int per = 0;
    if (Bn <= 0 || D <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0) return 0;
    const int s = halo_k_slices(Bn, D, H, W, Ci, Co, kd, &per);
    return s > 1 ? (size_t)s * Bn * D * H * W * Co * sizeof(float) : 0;
}
extern "C" int hupr_conv3x3_halo_bf16act_ws(const void* x, const void* wp_bf16, const float* bias, const void* res, void* y,
                                            int Bn, int D, int H, int W, int Ci, int in_ld, int Co, int out_ld, int res_ld,
                                            int kd, void* ws, size_t ws_bytes, hupr_stream_t stream) {
    return conv3x3_halo(x, wp_bf16, bias, res, y, Bn, D, H, W, Ci, in_ld, Co, out_ld, res_ld, kd, true, stream,
                        "hupr_conv3x3_halo_bf16act_ws", nullptr, ws, ws_bytes);
}

// Only the first half: the fp32 partial sums [slices][voxel][Co] stay in `part` (slices = hupr_conv3x3_halo_splitk_ws_bytes() /
// (voxels * Co * 4)) for a consumer that sums them itself (hupr_infer_tail_bf16act).  No bias, no residual, nothing stored to y.
extern "C" int hupr_conv3x3_halo_bf16act_partial(const void* x, const void* wp_bf16, int Bn, int D, int H, int W, int Ci, int in_ld,
                                                 int Co, int kd, void* part, size_t part_bytes, hupr_stream_t stream) {
    HUPR_REQUIRE(part, "hupr_conv3x3_halo_bf16act_partial: null pointer");
    return conv3x3_halo(x, wp_bf16, nullptr, nullptr, part /* never written: y of a partial-only launch */, Bn, D, H, W, Ci, in_ld, Co,
                        Co, 0, kd, true, stream, "hupr_conv3x3_halo_bf16act_partial", nullptr, part, part_bytes, true);
}

// BatchNorm statistics fused into the convolution (bf16 activations, 256-voxel kernel, no bias / residual): besides y the
// kernel leaves hupr_conv3x3_halo_stats_rows() rows of per-workgroup column sums [rows][2][Co] (doubles: sum, sum of
// squares of the STORED, i.e. bf16-rounded, outputs) in `stats`, which hupr_bn_train_finalize_f32 turns into the
// BatchNorm coefficients — the separate statistics pass over y (hupr_bn_train_stats_*) is not needed then.
extern "C" int hupr_conv3x3_halo_stats_supported(int Bn, int D, int H, int W, int Ci, int Co, int kd) {
    HaloArgs a{};
    a.kd = kd; a.D = D; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.in_ld = Ci;
    a.ablate = g_halo_ablate; a.trace = g_halo_trace;
    return (Bn > 0 && conv_halo256_stats_ok(a, Bn)) ? 1 : 0;
}
extern "C" int hupr_conv3x3_halo_stats_rows(void) { return kHalo256Grid; }
extern "C" int hupr_conv3x3_halo_bf16act_stats(const void* x, const void* wp_bf16, void* y, int Bn, int D, int H, int W, int Ci,
                                               int in_ld, int Co, int out_ld, int kd, void* stats, hupr_stream_t stream) {
    HUPR_REQUIRE(stats, "hupr_conv3x3_halo_bf16act_stats: null pointer");
    return conv3x3_halo(x, wp_bf16, nullptr, nullptr, y, Bn, D, H, W, Ci, in_ld, Co, out_ld, 0, kd, true, stream,
                        "hupr_conv3x3_halo_bf16act_stats", static_cast<double*>(stats));
}
