// 256-voxel persistent variant of the halo-tiled 3x3x3 / 1x3x3 convolution (bf16 matrix pipe).
//
// Phase ablation of the 128-voxel kernel on the dominant layer (scripts/halo_ablation.py; 64->64 3x3x3 at B=32,
// 385 us) showed the MFMA phase adding only ~107 us (near peak rate while it runs) on top of serialised per-workgroup
// overheads: weight re-streaming + barriers 98 us (221 KB of weights per 128 voxels — more than the halo itself),
// halo fill 67 us, output stores 50 us.  This variant attacks those:
//   * a 512-thread workgroup (8 waves = 4 x 64 voxels by 2 x 32 channels) owns a 4x8x8 (or 1x16x16) tile:
//     weights are streamed once per 256 voxels and the halo is 2.3x (not 3.1x) the tile;
//   * one persistent workgroup per CU walks a strided list of tiles; the first half of the NEXT tile's halo loads is
//     issued before the PREVIOUS tile's output stores, so loads and stores overlap in the memory system.
// Requirements: Ci % 64 == 0, Co % 64 == 0, and D % 4 == 0 (3-D) or H, W % 16 == 0 (2-D); otherwise the 128-voxel
// kernel is used.
#include "conv_halo.h"

namespace hupr {

constexpr int kHalo256MaxVox = 6 * 10 * 10;      // 3-D: (4+2) x 10 x 10 = 600;  2-D: 1 x 18 x 18 = 324

template <bool ABF>
__global__ __launch_bounds__(512) void hupr_k_conv_halo256_bf16(HaloArgs p) {
    constexpr int KC = 64, LDK = 64, BN = 64, TS = 3, C8 = 8;
    constexpr int NI = (kHalo256MaxVox * C8 + 511) / 512;      // 10 halo items (8 channels of a voxel) per thread ...
    constexpr int NH = ABF ? NI : NI / 2;                      // ... fp32 sources: two half batches (register budget: 256)
    constexpr int NB = TS * BN * C8 / 512;                     // 3 weight loads per thread per stage
    __shared__ __attribute__((aligned(16))) __bf16 Hs[kHalo256MaxVox * LDK];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[TS][BN * LDK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;                   // 4 x 2 waves, wave tile 64 voxels x 32 channels
    const int lr = lane & 31, lh = lane >> 5;
    const bool is3d = p.kd == 3;
    const int log2TW = is3d ? 3 : 4, log2TH = is3d ? 3 : 4;
    const int TW = 1 << log2TW, TH = 1 << log2TH, TD = is3d ? 4 : 1;
    const int pd = p.kd >> 1;
    const int HD = TD + p.kd - 1, HH = TH + 2, HW = TW + 2;
    const int T = p.kd * 9, n_stage = T / TS;
    const int nvox = HD * HH * HW;
    const int n_tiles = p.Bn * p.nd * p.nh * p.nw * p.n_co_tiles;
    const int n_chunks = p.Ci / KC;

    int abase[2], awx[2], ahy[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wm * 64 + i * 32 + lr;
        const int wx = row & (TW - 1), hy = (row >> log2TW) & (TH - 1), dz = row >> (log2TW + log2TH);
        abase[i] = ((dz * HH + hy) * HW + wx) * LDK;
        awx[i] = wx;
        ahy[i] = hy;
    }
    const int bkey = ((wn * 32 + lr) >> 1) & 7;

    // weight-stage loads of this thread: item f = tid + 512 j over [tap t][row n][chunk c8]
    int wt[NB], wdst[NB];
    long wsrc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int f = tid + 512 * j;
        const int t = f / (BN * C8), r = f % (BN * C8), n = r / C8, c8 = r % C8;
        wt[j] = t;
        wsrc[j] = (long)n * T * p.Ci + c8 * 8;
        wdst[j] = n * LDK + (((c8 ^ (n >> 1)) & 7) << 3);
    }

    f32x16 acc[2];
    u32x4 rb[NB];
    f32x4n va[ABF ? 1 : NH], vc[ABF ? 1 : NH];
    u32x4 vb[ABF ? NH : 1];
    int dst[NH];

#define HUPR_HALO_ISSUE(U0)                                                                                        \
    _Pragma("unroll") for (int u = 0; u < NH; ++u) {                                                                \
        const int it = tid + (u + (U0)) * 512;                                                                      \
        if constexpr (ABF) vb[u] = (u32x4){0u, 0u, 0u, 0u};                                                         \
        else { va[u] = (f32x4n){0.f, 0.f, 0.f, 0.f}; vc[u] = va[u]; }                                               \
        dst[u] = -1;                                                                                                \
        if (it < nvox * C8 && !(p.ablate & 1)) {                                                                    \
            const int vox = it >> 3, c8 = it & 7;                                                                   \
            const int hx = vox % HW;                                                                                \
            const int t_ = vox / HW;                                                                                \
            const int hy = t_ % HH, hz = t_ / HH;                                                                   \
            const int d = d0 + hz - pd, h = h0 + hy - 1, w = w0 + hx - 1;                                           \
            dst[u] = vox * LDK + ((c8 ^ (((hx >> 1) & 3) | ((hy & 1) << 2))) << 3);                                 \
            if ((unsigned)d < (unsigned)p.D && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W) {        \
                const long off = ((((long)b * p.D + d) * p.H + h) * p.W + w) * p.in_ld + c0 + c8 * 8;               \
                if constexpr (ABF) {                                                                                \
                    vb[u] = *reinterpret_cast<const u32x4*>(static_cast<const __bf16*>(p.x) + off);                 \
                } else {                                                                                            \
                    const float* src = static_cast<const float*>(p.x) + off;                                        \
                    va[u] = *reinterpret_cast<const f32x4n*>(src);                                                  \
                    vc[u] = *reinterpret_cast<const f32x4n*>(src + 4);                                              \
                }                                                                                                   \
            }                                                                                                       \
        }                                                                                                           \
    }
#define HUPR_HALO_COMMIT()                                                                                         \
    _Pragma("unroll") for (int u = 0; u < NH; ++u) {                                                                \
        if (dst[u] >= 0) {                                                                                          \
            if constexpr (ABF) {                                                                                    \
                *reinterpret_cast<u32x4*>(&Hs[dst[u]]) = vb[u];                                                     \
            } else {                                                                                                \
                bf16x8 v;                                                                                           \
                v[0] = (__bf16)va[u].x; v[1] = (__bf16)va[u].y; v[2] = (__bf16)va[u].z; v[3] = (__bf16)va[u].w;     \
                v[4] = (__bf16)vc[u].x; v[5] = (__bf16)vc[u].y; v[6] = (__bf16)vc[u].z; v[7] = (__bf16)vc[u].w;     \
                *reinterpret_cast<bf16x8*>(&Hs[dst[u]]) = v;                                                        \
            }                                                                                                       \
        }                                                                                                           \
    }
#define HUPR_STORE_TILE(B_, D0_, H0_, W0_, N0_)                                                                    \
    if (!(p.ablate & 4)) {                                                                                          \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                             \
            const int row = wm * 64 + i * 32 + lr;                                                                  \
            const int wx = row & (TW - 1), hy = (row >> log2TW) & (TH - 1), dz = row >> (log2TW + log2TH);          \
            const long m = (((long)(B_) * p.D + (D0_) + dz) * p.H + (H0_) + hy) * p.W + (W0_) + wx;                 \
            halo_store_voxel<ABF>(p, acc[i], m, (N0_) + wn * 32 + 4 * lh);                                          \
        }                                                                                                           \
    }

    // profiling: 8 s_memtime stamps per tile from wave 0 of workgroup 0 (all branches below are wave-uniform)
    const bool tracing = p.trace != nullptr && blockIdx.x == 0 && wave == 0;
    int tslot = 0;
#define HUPR_STAMP()                                                                            \
    if (tracing) {                                                                              \
        const unsigned long long t__ = __builtin_amdgcn_s_memtime();                            \
        if (lane == 0 && tslot < 4096) p.trace[tslot] = t__;                                    \
        ++tslot;                                                                                \
    }
    int pb = 0, pd0 = 0, ph0 = 0, pw0 = 0, pn0 = 0;            // previous tile (its accumulators are still live)
    bool have_prev = false;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        int b, d0, h0, w0, n0;
        {
            const int cot = tile % p.n_co_tiles;
            int st = tile / p.n_co_tiles;
            const int twi = st % p.nw; st /= p.nw;
            const int thi = st % p.nh; st /= p.nh;
            const int tdi = st % p.nd;
            b = st / p.nd;
            d0 = tdi * TD; h0 = thi * TH; w0 = twi * TW; n0 = cot * BN;
        }
        const __bf16* wbase = p.wp + (long)n0 * T * p.Ci;
        for (int ch = 0; ch < n_chunks; ++ch) {
            const int c0 = ch * KC;
            HUPR_STAMP()                                          // 0: tile/chunk start
            HUPR_HALO_ISSUE(0)                                    // first half of the halo loads in flight ...
            HUPR_STAMP()                                          // 1: halo loads issued
            if (ch == 0) {
                if (have_prev) HUPR_STORE_TILE(pb, pd0, ph0, pw0, pn0)   // ... while the previous tile is written out
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) rb[j] = *reinterpret_cast<const u32x4*>(wbase + wsrc[j] + (long)wt[j] * p.Ci + c0);
            HUPR_STAMP()                                          // 2: previous tile stored, stage-0 weights issued
            __syncthreads();                                     // every wave is done with Hs / Bs of the previous chunk
            HUPR_STAMP()                                          // 3: barrier passed
            HUPR_HALO_COMMIT()
            if constexpr (!ABF) {
                HUPR_HALO_ISSUE(NH)
                HUPR_HALO_COMMIT()
            }
            HUPR_STAMP()                                          // 4: halo committed to LDS
            for (int st_ = 0; st_ < n_stage; ++st_) {
#pragma unroll
                for (int j = 0; j < NB; ++j) *reinterpret_cast<u32x4*>(&Bs[wt[j]][wdst[j]]) = rb[j];
                __syncthreads();
                if (st_ == 0) { HUPR_STAMP() }                    // 5: first weight stage visible
                if (st_ + 1 < n_stage) {
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        rb[j] = *reinterpret_cast<const u32x4*>(wbase + wsrc[j] + (long)((st_ + 1) * TS + wt[j]) * p.Ci + c0);
                }
                if (!(p.ablate & 2)) {
#pragma unroll
                    for (int t = 0; t < TS; ++t) {
                        const int tap = st_ * TS + t;
                        const int tw_ = tap % 3, tt = tap / 3;
                        const int th_ = tt % 3, td_ = tt / 3;
                        const int toff = ((td_ * HH + th_) * HW + tw_) * LDK;
                        int akey[2];
#pragma unroll
                        for (int i = 0; i < 2; ++i) akey[i] = (((awx[i] + tw_) >> 1) & 3) | (((ahy[i] + th_) & 1) << 2);
#pragma unroll
                        for (int ks = 0; ks < KC / 16; ++ks) {
                            const int cw = ks * 2 + lh;
                            const bf16x8 bfrag = *reinterpret_cast<const bf16x8*>(&Bs[t][(wn * 32 + lr) * LDK + ((cw ^ bkey) << 3)]);
#pragma unroll
                            for (int i = 0; i < 2; ++i) {
                                const bf16x8 afrag = *reinterpret_cast<const bf16x8*>(&Hs[abase[i] + toff + ((cw ^ akey[i]) << 3)]);
                                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfrag, afrag, acc[i], 0, 0, 0);   // D'[channel][voxel]
                            }
                        }
                    }
                }
                __syncthreads();
                if (st_ == 0) { HUPR_STAMP() }                    // 6: first stage computed
            }
            HUPR_STAMP()                                          // 7: all stages done
        }
        pb = b; pd0 = d0; ph0 = h0; pw0 = w0; pn0 = n0;
        have_prev = true;
    }
    if (have_prev) HUPR_STORE_TILE(pb, pd0, ph0, pw0, pn0)
#undef HUPR_STAMP
#undef HUPR_HALO_ISSUE
#undef HUPR_HALO_COMMIT
#undef HUPR_STORE_TILE
}

bool launch_conv_halo256(HaloArgs a, int Bn, bool abf, hipStream_t s) {
    const bool big3 = (a.kd == 3 && a.D % 4 == 0 && a.H % 8 == 0 && a.W % 8 == 0);
    const bool big2 = (a.kd == 1 && a.D == 1 && a.H % 16 == 0 && a.W % 16 == 0);
    // measured (scripts/halo_ablation.py): +8 % on the 3-D encoder layers, neutral to -6 % on the 2-D decoder maps
    (void)big2;
    if (a.Ci % 64 != 0 || a.Co % 64 != 0 || !big3) return false;
    a.TD = big3 ? 4 : 1;
    a.log2TW = big3 ? 3 : 4;
    a.nd = a.D / a.TD;
    a.nh = a.H / (big3 ? 8 : 16);
    a.nw = a.W >> a.log2TW;
    a.n_co_tiles = a.Co / 64;
    const long tiles = (long)Bn * a.nd * a.nh * a.nw * a.n_co_tiles;
    if (tiles >= (1L << 31) || tiles < 256) return false;          // small problems: the 128-voxel kernel fills the chip better
    // one persistent workgroup per CU
    if (abf) hipLaunchKernelGGL(hupr_k_conv_halo256_bf16<true>, dim3(256), dim3(512), 0, s, a);
    else hipLaunchKernelGGL(hupr_k_conv_halo256_bf16<false>, dim3(256), dim3(512), 0, s, a);
    return true;
}

}  // namespace hupr
