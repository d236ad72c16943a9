// Streaming temporal merge (Conv3d kernel (G,1,1), reference models/layers.py:208-210,218-220) for the level-1 maps:
// merged[b][v][co] = sum_g sum_ci x[b][g][v][ci] * W[co][ci][g],  x bf16 (B, G, HW, 64), merged fp32 (B, HW, 64).
//
// The generic bf16 GEMM engine keeps ONE K-tile (8 KB) of loads in flight per workgroup behind its MFMAs; at 2-4 workgroups per
// CU that is 4-8 MB chip-wide against the ~16 MB that 8 TB/s x 2 us of latency needs, and the merge — 64 flop/B, far below the
// ridge — ran at 2.2 TB/s (77 us for 168 MB).  Here the operand travels global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds:
// no staging registers) through a ring of four 16 KB stages per workgroup: a stage = one frame slice of a 128-voxel tile, which
// is ONE contiguous 16 KB block of x; three stages (48 KB) are in flight per CU while the fourth is multiplied.  The eight
// 64 x 64 weight slices stay in LDS for the whole launch (persistent workgroups).  D' = W X^T as in the halo convolution:
// a lane ends up with one voxel and 4 x 4 consecutive channels (16-byte stores).
#include "conv_halo.h"

namespace hupr {

constexpr int kTmStages = 4;

template <int G>
__global__ __launch_bounds__(256) void hupr_k_tmerge_fwd_stream(const __bf16* __restrict__ x, const __bf16* __restrict__ wp,
                                                                float* __restrict__ y, int Bn, int HW) {
    constexpr int C = 64, ROW = C;                                // bf16 elements per LDS row (128 B, 16-byte chunks swizzled)
    __shared__ __attribute__((aligned(16))) __bf16 Ws[G][C * ROW];               // [g][co][ci]
    __shared__ __attribute__((aligned(16))) __bf16 As[kTmStages][128 * ROW];     // [slot][voxel][ci]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;

    // weights: packed [co][g][ci] bf16 -> Ws[g][co][chunk ^ ((co >> 1) & 7)]
    for (int it = tid; it < G * C * 8; it += 256) {
        const int c8 = it & 7, co = (it >> 3) % C, g = it / (8 * C);
        *reinterpret_cast<u32x4*>(&Ws[g][co * ROW + ((c8 ^ ((co >> 1) & 7)) << 3)]) =
            *reinterpret_cast<const u32x4*>(wp + ((long)co * G + g) * C + c8 * 8);
    }

    const int tiles_per_b = HW / 128, n_tiles = Bn * tiles_per_b;
    const int my_tiles = ((int)blockIdx.x < n_tiles) ? (n_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int n_stage = my_tiles * G;
    if (n_stage == 0) return;

    // LDS-DMA of stage n (tile blockIdx.x + (n / G) * gridDim.x, frame n % G) into slot n % 4: 16 pieces of 1 KiB = 8 rows; wave w
    // moves pieces 4 w .. 4 w + 3; lane l deposits 16 bytes at piece base + 16 l = row (l >> 3), chunk position l & 7, and the
    // row swizzle is applied on the SOURCE side (position c' holds source chunk c' ^ key(row)).  Inline asm with M0 saved /
    // restored: hipcc neither counts these loads nor guards later LDS reads with vmcnt(0); the waits below are explicit.
    const u32x4 xrs = {(unsigned)(unsigned long)x, (unsigned)((unsigned long)x >> 32) & 0xffffu,
                       (unsigned)((long)Bn * G * HW * C * 2), 0x00020000u};
    const unsigned as_lds = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)&As[0][0];
    auto dma = [&](int n) {
        const int t = (int)blockIdx.x + (n / G) * (int)gridDim.x, g = n % G;
        const int b = t / tiles_per_b, v0 = (t % tiles_per_b) * 128;
        const unsigned src0 = (unsigned)((((long)b * G + g) * HW + v0) * C * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (4 * wave + j) * 8 + (lane >> 3);
            const unsigned voff = src0 + row * (C * 2) + ((((lane & 7) ^ ((row >> 1) & 7)) & 7) << 4);
            const unsigned dst = __builtin_amdgcn_readfirstlane(as_lds + (n % kTmStages) * (128 * ROW * 2) + (4 * wave + j) * 1024);
            unsigned keep_;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep_)
                         : "s"(dst), "v"(voff), "s"(xrs)
                         : "memory");
        }
    };
#define HUPR_VMCNT(N_) __builtin_amdgcn_s_waitcnt(0x0F70 | ((N_) & 15) | (((N_) >> 4) << 14))

    for (int n = 0; n < kTmStages - 1 && n < n_stage; ++n) dma(n);

    f32x16 acc[2];
    const int arow = (wave * 32 + lr) * ROW, akey = ((wave * 32 + lr) >> 1) & 7;
    bool stored = false;                                         // did the previous iteration end with a tile's 8 stores?
    for (int n = 0; n < n_stage; ++n) {
        const int g = n % G;
        // stage n has landed: only the (up to two) younger stages, and a just-finished tile's stores, may still be in flight
        const int younger = min(kTmStages - 2, n_stage - 1 - n) * 4;
        if (stored) { if (younger >= 8) HUPR_VMCNT(16); else if (younger == 4) HUPR_VMCNT(12); else HUPR_VMCNT(8); }
        else { if (younger >= 8) HUPR_VMCNT(8); else if (younger == 4) HUPR_VMCNT(4); else HUPR_VMCNT(0); }
        __syncthreads();                                          // everyone's pieces of stage n are in LDS; slot (n - 1) % 4 is free
        if (n + kTmStages - 1 < n_stage) dma(n + kTmStages - 1);
        if (g == 0) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
        }
        const __bf16* At = &As[n % kTmStages][0];
        const __bf16* Wt = &Ws[g][0];
#pragma unroll
        for (int ks = 0; ks < C / 16; ++ks) {
            const int cw = ks * 2 + lh;
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(&At[arow + ((cw ^ akey) << 3)]);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int co = 32 * ct + lr;
                const bf16x8 bq = *reinterpret_cast<const bf16x8*>(&Wt[co * ROW + ((cw ^ ((co >> 1) & 7)) << 3)]);
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq, af, acc[ct], 0, 0, 0);
            }
        }
        stored = false;
        if (g == G - 1) {
            const int t = (int)blockIdx.x + (n / G) * (int)gridDim.x;
            const long m = (long)(t / tiles_per_b) * HW + (t % tiles_per_b) * 128 + wave * 32 + lr;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
                    *reinterpret_cast<f32x4n*>(y + m * C + 32 * ct + 8 * q4 + 4 * lh) =
                        (f32x4n){acc[ct][4 * q4], acc[ct][4 * q4 + 1], acc[ct][4 * q4 + 2], acc[ct][4 * q4 + 3]};
            stored = true;
        }
    }
#undef HUPR_VMCNT
}

}  // namespace hupr

using namespace hupr;

extern "C" int hupr_tmerge_stream_supported(int G, int HW, int Ci, int Co) {
    return (Ci == 64 && Co == 64 && HW % 128 == 0 && (G == 8 || G == 4 || G == 2)) ? 1 : 0;
}

// x bf16 (Bn, G, HW, 64) channels-last; wp_bf16: the merge weight packed [Co][G][Ci] (hupr_pack_conv_weights_bf16 mode 0);
// y fp32 (Bn, HW, 64).  Stream-ordered, no workspace.
extern "C" int hupr_tmerge_fwd_stream_bf16(const void* x, const void* wp_bf16, float* y, int Bn, int G, int HW, int Ci, int Co,
                                           hupr_stream_t stream) {
    HUPR_REQUIRE(x && wp_bf16 && y && Bn > 0, "hupr_tmerge_fwd_stream_bf16: bad argument");
    HUPR_REQUIRE(hupr_tmerge_stream_supported(G, HW, Ci, Co), "hupr_tmerge_fwd_stream_bf16: unsupported geometry (G=%d HW=%d Ci=%d Co=%d)", G, HW, Ci, Co);
    HUPR_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)wp_bf16 & 15) == 0 && ((uintptr_t)y & 15) == 0, "hupr_tmerge_fwd_stream_bf16: misaligned pointer");
    HUPR_REQUIRE((long)Bn * G * HW * Ci * 2 < 0x7fffffffL, "hupr_tmerge_fwd_stream_bf16: tensor too large for 32-bit buffer offsets");
    const int tiles = Bn * (HW / 128);
    const dim3 grid((unsigned)min(tiles, 256));
    const __bf16* xb = static_cast<const __bf16*>(x);
    const __bf16* wb = static_cast<const __bf16*>(wp_bf16);
    hipStream_t s = as_stream(stream);
    if (G == 8) hipLaunchKernelGGL(hupr_k_tmerge_fwd_stream<8>, grid, dim3(256), 0, s, xb, wb, y, Bn, HW);
    else if (G == 4) hipLaunchKernelGGL(hupr_k_tmerge_fwd_stream<4>, grid, dim3(256), 0, s, xb, wb, y, Bn, HW);
    else hipLaunchKernelGGL(hupr_k_tmerge_fwd_stream<2>, grid, dim3(256), 0, s, xb, wb, y, Bn, HW);
    HUPR_LAUNCH_OK("hupr_k_tmerge_fwd_stream");
    return HUPR_OK;
}
