// 512-voxel register-blocked variant of the halo-tiled 3x3x3 convolution (bf16 matrix pipe, bf16 activations) for the 3-D
// encoder layers — the successor of the 256-voxel kernel for the shapes that dominate the step.
//
// What the 256-voxel kernel's phase ablations said (scripts/halo_ablation.py, layer 1 at B = 32): with every other phase
// removed the MFMA stream alone takes 146 us (the matrix pipe sustains ~1 850 TF/s on random operands at this chip's
// power limit, scripts/probes/mfma_peak_probe.hip), fragment re-reads cost +17 us, weight staging +11 us, the epilogue +9 us.
// This kernel goes after the first two by doing more MFMAs per byte moved through LDS:
//   * a wave owns 2 depth slices x 2 rows x 32 columns (8 x, 4 row pairs) x 32 channels = FOUR accumulator tiles (the 256-voxel
//     kernel: two).  A stage is one kx column with all nine (kz, ky) taps; walking kz, the depth rows of the halo are
//     reused by the next depth tap exactly as the ky taps reuse the y rows: per K-step 16 halo fragments + 9 weight fragments
//     feed 36 MFMAs (0.69 reads per MFMA instead of 1.17);
//   * the reduction axis is split in chunks of 32 input channels: the halo of a 4 x 8 x 16 tile is then 69 KB, a stage's nine
//     taps 36 KB (double-buffered, LDS-DMA), and a barrier is needed once per 72 MFMAs of a wave instead of once per 24;
//   * accumulators persist across the Ci / 32 chunks of a tile; Ci = 32 (the encoder's first layer) is one chunk.
// Envelope: kd = 3, D % 4 == 0, H % 8 == 0, W % 16 == 0, Ci % 32 == 0, Co % 64 == 0, bf16 activations, >= 256 tiles;
// bias / residual in the immediate epilogue; anything else falls through to the 256- / 128-voxel kernels.
//
// MEASURED (scripts/conv_variants_ab.py, profiles/r02_conv_variants_ab.txt): the tap loop is NOT faster than the 256-voxel
// kernel's (skeleton 176 vs 180 us on layer 1) although it reads 41 % fewer fragment bytes and crosses a third of the
// barriers — the matrix pipe fed from LDS is power-limited well below its nominal rate (scripts/probes/mfma_peak_probe.hip:
// 1 850 TF/s register-only, 1 475 / 1 573 TF/s with 7 / 4 fragment reads per 6 MFMAs), and the bigger tile pays more for
// fill and epilogue.  It wins only where the 256-voxel kernel does not apply: Ci = 32 (the encoder's first layer, one chunk:
// 130 vs 150 us).  The dispatcher therefore selects it for Ci == 32 only.
#include "conv_halo.h"

namespace hupr {

__global__ __launch_bounds__(512) void hupr_k_conv_halo512_bf16(HaloArgs p) {
    constexpr int KC = 32, LDK = 32, BN = 64;                   // channels per chunk = bf16 elements per LDS row (64 B)
    constexpr int TD = 4, TH = 8, TW = 16, HD = TD + 2, HH = TH + 2, HW = TW + 2;
    constexpr int NVOX = HD * HH * HW;                         // 1080 halo voxels
    constexpr int T = 27;
    constexpr int NI = (NVOX * 4 + 511) / 512;                 // 9 halo items (8 channels = 16 B of a voxel) per thread
    constexpr int STAGE_B = 9 * BN * LDK * 2;                  // 36 864 B: nine taps x 64 rows x 64 B
    __shared__ __attribute__((aligned(16))) __bf16 Hs[NVOX * LDK];
    __shared__ __attribute__((aligned(1024))) __bf16 Bs[2][9][BN * LDK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 1, zp = (wave >> 1) & 1, xh = wave >> 2;
    const int lr = lane & 31, lh = lane >> 5;
    const int wx = lr & 7, yp = lr >> 3;
    // halo element of this lane's voxel column for (depth row 0, y row 0, kx 0)
    const int abase = (((2 * zp) * HH + 2 * yp) * HW + 8 * xh + wx) * LDK;
    // row swizzle: 16-byte chunk c (0..3) of halo voxel (hy, hx) lives at chunk c ^ ((((hy >> 1) & 1) << 1) | ((hx >> 2) & 1)):
    // with 64-byte rows every 16-lane ds_read_b128 group (lane groups per MI355X_MICROARCH.md) covers all 64 banks for every tap
    const int e0 = yp & 1;
    int xk[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) xk[kx] = ((8 * xh + wx + kx) >> 2) & 1;
    // weights: chunk c of row n lives at chunk c ^ ((n >> 2) & 3)
    const int brow = wn * 32 + lr;
    const int bbase = brow * LDK;
    const int wkey = (brow >> 2) & 3;
    const int n_tiles = p.Bn * p.nd * p.nh * p.nw * p.n_co_tiles;

    // ---- weight stages by LDS-DMA: 36 pieces of 1 KiB (16 rows of one tap); wave w moves pieces w, w + 8, ... ---------
    const int wrow_ = lane >> 2;                                 // row inside a piece
    const int wchk_ = lane & 3;                                  // chunk POSITION inside the row
    const u32x4 wrs = {(unsigned)(unsigned long)p.wp, (unsigned)((unsigned long)p.wp >> 32) & 0xffffu,
                       (unsigned)((long)p.Co * T * p.Ci * 2), 0x00020000u};
    const unsigned bs_lds = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)&Bs[0][0][0];
#define HUPR_W_DMA(COT_, CH_, KX_, PAR_)                                                                            \
    _Pragma("unroll") for (int j = 0; j < 5; ++j) {                                                                 \
        const int pc_ = wave + 8 * j;                            /* wave-uniform */                                 \
        if (pc_ < 36) {                                                                                             \
            const int t9_ = pc_ >> 2;                            /* (kz, ky) */                                     \
            const int n_ = (pc_ & 3) * 16 + wrow_;                                                                  \
            const int tap_ = t9_ * 3 + (KX_);                                                                       \
            const int src_ = ((((COT_) * BN + n_) * T + tap_) * p.Ci + (CH_) * KC + ((wchk_ ^ (n_ >> 2)) & 3) * 8) * 2; \
            const unsigned dst_ = __builtin_amdgcn_readfirstlane(bs_lds + (PAR_) * STAGE_B + pc_ * 1024);            \
            unsigned keep_;                                                                                         \
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\t" \
                         "s_mov_b32 m0, %0"                                                                         \
                         : "=&s"(keep_) : "s"(dst_), "v"(src_), "s"(wrs) : "memory");                               \
        }                                                                                                           \
    }
#define HUPR_VMCNT(N_) __builtin_amdgcn_s_waitcnt(0x0F70 | ((N_) & 15) | (((N_) >> 4) << 14))

    // ---- halo: global -> registers (raw buffer loads, out-of-range = zero padding) -> LDS -----------------------------
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.x), 0, (int)((long)p.Bn * p.D * p.H * p.W * p.in_ld * 2), 0x00020000);
    u32x4 vb[NI];
    // (tid is laundered through an empty asm in both macros: hipcc otherwise hoists the ~30 tile-invariant per-item offsets
    // out of the item loop and keeps them in VGPRs for the whole kernel — this kernel has none to spare)
#define HUPR_HALO_ISSUE(B_, D0_, H0_, W0_, C0_)                                                                     \
    int tid_i_ = tid;                                                                                               \
    asm volatile("" : "+v"(tid_i_));                                                                                \
    _Pragma("unroll") for (int u = 0; u < NI; ++u) {                                                                \
        const int it = tid_i_ + u * 512;                                                                            \
        const int vox = it >> 2, c8 = it & 3;                                                                       \
        const int hx = vox % HW;                                                                                    \
        const int t_ = vox / HW;                                                                                    \
        const int hy = t_ % HH, hz = t_ / HH;                                                                       \
        const int d = (D0_) + hz - 1, h = (H0_) + hy - 1, w = (W0_) + hx - 1;                                       \
        const bool ok = it < NVOX * 4 && !(p.ablate & 1) && (unsigned)d < (unsigned)p.D &&                          \
                        (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;                                 \
        const int off = (((((B_) * p.D + d) * p.H + h) * p.W + w) * p.in_ld + (C0_) + c8 * 8) * 2;                  \
        const auto ld_ = __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? off : 0x7ffffff0, 0, 0);                   \
        vb[u] = (u32x4){ld_[0], ld_[1], ld_[2], ld_[3]};                                                            \
    }
#define HUPR_HALO_COMMIT()                                                                                          \
    int tid_c_ = tid;                                                                                               \
    asm volatile("" : "+v"(tid_c_));                                                                                \
    _Pragma("unroll") for (int u = 0; u < NI; ++u) {                                                                \
        const int it = tid_c_ + u * 512;                                                                            \
        if (it < NVOX * 4 && !(p.ablate & 1)) {                                                                     \
            const int vox = it >> 2, c8 = it & 3;                                                                   \
            const int hx = vox % HW, hy = (vox / HW) % HH;                                                          \
            *reinterpret_cast<u32x4*>(&Hs[vox * LDK + ((c8 ^ ((((hy >> 1) & 1) << 1) | ((hx >> 2) & 1))) << 3)]) = vb[u]; \
        }                                                                                                           \
    }

    // ---- work items = (tile, channel chunk): contiguous tile range per workgroup, XCD-aware (see the 256-voxel kernel) ----
    struct Pos { int cot, twi, thi, tdi, b, ch; };
    const int n_chunks = p.Ci / KC;
    const int per_wg = (n_tiles + gridDim.x - 1) / gridDim.x;
    const int wg_rank = (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));
    const int t_begin = wg_rank * per_wg, t_end = min(n_tiles, t_begin + per_wg);
    if (t_begin >= t_end) return;
    Pos cur;
    {
        cur.cot = t_begin % p.n_co_tiles;
        int st = t_begin / p.n_co_tiles;
        cur.twi = st % p.nw; st /= p.nw;
        cur.thi = st % p.nh; st /= p.nh;
        cur.tdi = st % p.nd;
        cur.b = st / p.nd;
        cur.ch = 0;
    }
    const int n_items = (t_end - t_begin) * n_chunks;

    f32x16 acc[2][2];                                             // [zi][yi]
    HUPR_W_DMA(cur.cot, cur.ch, 0, 0)
    { HUPR_HALO_ISSUE(cur.b, cur.tdi * TD, cur.thi * TH, cur.twi * TW, cur.ch * KC) }
    { HUPR_HALO_COMMIT() }
    HUPR_VMCNT(0);
    __syncthreads();

    int g = 0;                                                    // global stage counter: stage g reads Bs[g & 1]
    for (int q = 0; q < n_items; ++q) {
        const int b = cur.b, d0 = cur.tdi * TD, h0 = cur.thi * TH, w0 = cur.twi * TW, n0 = cur.cot * BN;
        const bool first_chunk = cur.ch == 0, last_chunk = cur.ch == n_chunks - 1;
        Pos nxt = cur;
        if (++nxt.ch == n_chunks) {
            nxt.ch = 0;
            if (++nxt.cot == p.n_co_tiles) {
                nxt.cot = 0;
                if (++nxt.twi == p.nw) {
                    nxt.twi = 0;
                    if (++nxt.thi == p.nh) {
                        nxt.thi = 0;
                        if (++nxt.tdi == p.nd) { nxt.tdi = 0; ++nxt.b; }
                    }
                }
            }
        }
        const bool has_next = q + 1 < n_items;
        if (first_chunk) {
#pragma unroll
            for (int zi = 0; zi < 2; ++zi)
#pragma unroll
                for (int yi = 0; yi < 2; ++yi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[zi][yi][r] = 0.f;
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int par = (g + kx) & 1;
            // next stage's nine taps -> the idle half of Bs (everyone left it at the previous barrier)
            if (kx + 1 < 3) { HUPR_W_DMA(cur.cot, cur.ch, kx + 1, par ^ 1) }
            else if (has_next) { HUPR_W_DMA(nxt.cot, nxt.ch, 0, par ^ 1) }
            if (kx == 0 && has_next) {                            // next item's halo rides under this item's stages
                { HUPR_HALO_ISSUE(nxt.b, nxt.tdi * TD, nxt.thi * TH, nxt.twi * TW, nxt.ch * KC) }
            }
            {
                const __bf16* Bt = &Bs[par][0][0];
                // fragments: halo rows R[zr 0..3][yr 0..3] of this lane's column, weights W[kz][ky]; one (K-step, kz) group =
                // 4 new halo fragments (depth row kz + 1; for kz = 0 also row 0) + 3 weight fragments -> 12 MFMAs
                bf16x8 R[4][4], Wf[2][3];
#define HUPR_A_ROW(ZR_, KS_)                                                                                        \
                _Pragma("unroll") for (int yr = 0; yr < 4; ++yr) {                                                  \
                    const int cst_ = (KS_) ^ (yr >> 1);                                                             \
                    const int chunk_ = (((e0 ^ cst_) << 1) | (lh ^ xk[kx]));                                        \
                    R[ZR_][yr] = *reinterpret_cast<const bf16x8*>(                                                  \
                        &Hs[abase + (((ZR_) * HH + yr) * HW + kx) * LDK + (chunk_ << 3)]);                           \
                }
#define HUPR_B_ROW(SET_, KZ_, KS_)                                                                                  \
                _Pragma("unroll") for (int ky = 0; ky < 3; ++ky)                                                    \
                    Wf[SET_][ky] = *reinterpret_cast<const bf16x8*>(                                                \
                        &Bt[((KZ_) * 3 + ky) * (BN * LDK) + bbase + ((((KS_) * 2 + lh) ^ wkey) << 3)]);
#define HUPR_MMA(KZ_, SET_)                                                                                         \
                _Pragma("unroll") for (int ky = 0; ky < 3; ++ky)                                                    \
                    _Pragma("unroll") for (int zi = 0; zi < 2; ++zi)                                                \
                        _Pragma("unroll") for (int yi = 0; yi < 2; ++yi)                                            \
                            acc[zi][yi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wf[SET_][ky], R[(KZ_) + zi][yi + ky], acc[zi][yi], 0, 0, 0);
                HUPR_A_ROW(0, 0) HUPR_A_ROW(1, 0) HUPR_B_ROW(0, 0, 0)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int s0 = ks & 1;                        // weight-fragment set that holds (K-step ks, kz = 0)
                    // kz = 0: prefetch depth row 2 + weights of kz = 1
                    // (the prefetch reads of a group are spread in front of the MFMAs of the previous one instead of issued as a
                    // burst, see conv_halo256m_bf16.hip)
#define HUPR_SPREAD(NR_)                                                                                            \
                    _Pragma("unroll") for (int i_ = 0; i_ < (NR_); ++i_) {                                          \
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                          \
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                          \
                    }                                                                                               \
                    __builtin_amdgcn_sched_group_barrier(0x008, 12 - (NR_), 0);                                     \
                    __builtin_amdgcn_sched_barrier(0);
                    HUPR_A_ROW(2, ks) HUPR_B_ROW(s0 ^ 1, 1, ks)
                    HUPR_MMA(0, s0)
                    HUPR_SPREAD(7)
                    // kz = 1: prefetch depth row 3 + weights of kz = 2
                    HUPR_A_ROW(3, ks) HUPR_B_ROW(s0, 2, ks)
                    HUPR_MMA(1, s0 ^ 1)
                    HUPR_SPREAD(7)
                    // kz = 2: prefetch the next K-step's depth rows 0, 1 (dead by now) + its weights of kz = 0
                    if (ks == 0) { HUPR_A_ROW(0, 1) HUPR_A_ROW(1, 1) HUPR_B_ROW(s0 ^ 1, 0, 1) }
                    HUPR_MMA(2, s0)
                    HUPR_SPREAD(11)
#undef HUPR_SPREAD
                }
#undef HUPR_A_ROW
#undef HUPR_B_ROW
#undef HUPR_MMA
            }
            // this wave's pieces of the next stage have landed (stage 0: the next halo's NI younger register loads stay in flight)
            if (kx == 0 && has_next) { HUPR_VMCNT(NI); } else { HUPR_VMCNT(0); }
            __syncthreads();                                      // Bs[par ^ 1] complete; all waves done with Bs[par] (after kx = 2: with Hs)
        }
        g += 3;
        if (last_chunk && !(p.ablate & 4)) {
#pragma unroll
            for (int zi = 0; zi < 2; ++zi)
#pragma unroll
                for (int yi = 0; yi < 2; ++yi) {
                    const long m = (((long)b * p.D + d0 + 2 * zp + zi) * p.H + h0 + 2 * yp + yi) * p.W + w0 + 8 * xh + wx;
                    halo_store_voxel<true>(p, acc[zi][yi], m, n0 + wn * 32 + 4 * lh);
                }
        }
        if (has_next) {
            { HUPR_HALO_COMMIT() }
            __syncthreads();
        }
        cur = nxt;
    }
#undef HUPR_W_DMA
#undef HUPR_VMCNT
#undef HUPR_HALO_ISSUE
#undef HUPR_HALO_COMMIT
}

bool conv_halo512_supported(const HaloArgs& a, int Bn, bool abf) {
    if (!abf || a.kd != 3 || a.D % 4 != 0 || a.H % 8 != 0 || a.W % 16 != 0 || a.Ci % 32 != 0 || a.Co % 64 != 0 || a.stats) return false;
    if ((a.Co & 7) || (a.out_ld & 7) || (a.in_ld & 7)) return false;
    const long tiles = (long)Bn * (a.D / 4) * (a.H / 8) * (a.W / 16) * (a.Co / 64);
    if (tiles >= (1L << 31) || tiles < 256) return false;
    if ((long)Bn * a.D * a.H * a.W * a.in_ld * 2 >= 0x7ffffff0L || (long)a.Co * 27 * a.Ci * 2 >= 0x7ffffff0L) return false;
    return true;
}

bool launch_conv_halo512(HaloArgs a, int Bn, bool abf, hipStream_t s) {
    if (!conv_halo512_supported(a, Bn, abf)) return false;
    if (a.Ci != 32) return false;                               // see the measurement note at the top of this file
    a.TD = 4;
    a.log2TW = 4;
    a.nd = a.D / 4;
    a.nh = a.H / 8;
    a.nw = a.W / 16;
    a.n_co_tiles = a.Co / 64;
    HUPR_LAUNCH(hupr_k_conv_halo512_bf16, dim3(kHalo256Grid), dim3(512), 0, s, a);      // one persistent workgroup per CU
    return true;
}

}  // namespace hupr
