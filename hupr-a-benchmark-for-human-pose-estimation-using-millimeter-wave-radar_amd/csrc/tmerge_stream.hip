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
//
// This file: the forward kernel (below), the write-bound input gradient (hupr_k_tmerge_dgrad_stream) and the weight gradient with both
// operands transposed on their way out of LDS (hupr_k_tmerge_wgrad_stream, which also serves the 128- / 256-channel merges of levels 2
// and 3 through "virtual" 64-channel frames) with its fixed-order partial reduction.  In the step: 41 / 41 / 42 + 7 us per level-1
// call against 77 / 79 / 102 us on the generic engine ("Streaming temporal merges").
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


// ---- input gradient: dx[b][g][v][ci] = sum_co dy[b][v][co] * W[co][ci][g] -------------------------------------------------------------
// Write-bound (134 MB of bf16 dx against 34 MB of fp32 dy at B = 32): a persistent workgroup converts a 128-voxel dy tile to bf16 in
// LDS once (the next tile's 32 KB already on their way into registers), multiplies it by the G resident weight slices
// (D' = W_g^T dy^T: a lane holds one voxel and 4 x 4 input channels) and stores G x 16 KB with 16-byte stores (lanes l / l + 32
// exchange their 4-channel halves, as in the halo convolution's bf16 epilogue).
template <int G>
__global__ __launch_bounds__(256) void hupr_k_tmerge_dgrad_stream(const float* __restrict__ dy, const __bf16* __restrict__ wp1,
                                                                  __bf16* __restrict__ dx, int Bn, int HW) {
    constexpr int C = 64, ROW = C;
    __shared__ __attribute__((aligned(16))) __bf16 Wt[G][C * ROW];               // [g][ci][co]
    __shared__ __attribute__((aligned(16))) __bf16 Ds[2][128 * ROW];             // [buffer][voxel][co]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;

    // weights: mode-1 packed [ci][G - 1 - g][co] bf16 -> Wt[g][ci][chunk ^ ((ci >> 1) & 7)]
    for (int it = tid; it < G * C * 8; it += 256) {
        const int c8 = it & 7, ci = (it >> 3) % C, g = it / (8 * C);
        *reinterpret_cast<u32x4*>(&Wt[g][ci * ROW + ((c8 ^ ((ci >> 1) & 7)) << 3)]) =
            *reinterpret_cast<const u32x4*>(wp1 + ((long)ci * G + (G - 1 - g)) * C + c8 * 8);
    }

    const int tiles_per_b = HW / 128, n_tiles = Bn * tiles_per_b;
    const int my_tiles = ((int)blockIdx.x < n_tiles) ? (n_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    if (my_tiles == 0) return;

    f32x4n pre[8];                                                // a dy tile: 128 x 64 fp32 = 8 float4 per thread, contiguous in HBM
    auto fetch = [&](int i) {
        const int t = (int)blockIdx.x + i * (int)gridDim.x;
        const float* src = dy + ((long)(t / tiles_per_b) * HW + (long)(t % tiles_per_b) * 128) * C;
#pragma unroll
        for (int j = 0; j < 8; ++j) pre[j] = *reinterpret_cast<const f32x4n*>(src + (long)(j * 256 + tid) * 4);
    };
    fetch(0);
    const int arow = (wave * 32 + lr) * ROW, akey = ((wave * 32 + lr) >> 1) & 7;
    for (int i = 0; i < my_tiles; ++i) {
        __bf16* Dt = &Ds[i & 1][0];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int f = j * 256 + tid, row = f >> 4, c4 = f & 15;
            const bf16x4 o = {(__bf16)pre[j][0], (__bf16)pre[j][1], (__bf16)pre[j][2], (__bf16)pre[j][3]};
            *reinterpret_cast<bf16x4*>(&Dt[row * ROW + (((c4 >> 1) ^ ((row >> 1) & 7)) << 3) + ((c4 & 1) << 2)]) = o;
        }
        if (i + 1 < my_tiles) fetch(i + 1);
        __syncthreads();                                          // buffer i & 1 complete; buffer (i + 1) & 1 was last read two tiles ago
        const int t = (int)blockIdx.x + i * (int)gridDim.x;
        const int b = t / tiles_per_b, v = (t % tiles_per_b) * 128 + wave * 32 + lr;
        bf16x8 af[C / 16];
#pragma unroll
        for (int ks = 0; ks < C / 16; ++ks) af[ks] = *reinterpret_cast<const bf16x8*>(&Dt[arow + (((ks * 2 + lh) ^ akey) << 3)]);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            f32x16 acc[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
            const __bf16* Wg = &Wt[g][0];
#pragma unroll
            for (int ks = 0; ks < C / 16; ++ks) {
                const int cw = ks * 2 + lh;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const int ci = 32 * ct + lr;
                    const bf16x8 bq = *reinterpret_cast<const bf16x8*>(&Wg[ci * ROW + ((cw ^ ((ci >> 1) & 7)) << 3)]);
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq, af[ks], acc[ct], 0, 0, 0);
                }
            }
            __bf16* drow = dx + (((long)b * G + g) * HW + v) * C;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                unsigned pk[8];
                halo_pack_tile(acc[ct], pk);
#pragma unroll
                for (int gp = 0; gp < 4; gp += 2) {
                    const auto r0 = __builtin_amdgcn_permlane32_swap(pk[2 * gp], pk[2 * gp + 2], false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(pk[2 * gp + 1], pk[2 * gp + 3], false, false);
                    *reinterpret_cast<u32x4*>(drow + 32 * ct + 8 * (gp + lh)) = (u32x4){r0[0], r1[0], r0[1], r1[1]};
                }
            }
        }
    }
}

// ---- weight gradient: dW[co][ci][g] = sum_{b,v} dy[b][v][co] * x[b][g][v][ci] ------------------------------------------------------------
// The reduction axis (voxels) has to run along the MFMA K of BOTH operands: ds_read_b64_tr_b16 produces those transposed fragments
// from the row-major [voxel][channel] LDS images (the idiom of wgrad_halo_bf16.hip).  Everything arrives by LDS-DMA through the
// same ring of four 16 KB stages as the forward kernel: per 128-voxel tile two stages carry the fp32 dy tile (converted LDS -> LDS
// to one bf16 image, its eight K-step fragments then live in registers for the tile) and G stages the frame slices of x.  Wave w
// owns the 32 x 32 block (co half w >> 1, ci half w & 1) of every frame's 64 x 64 gradient: G accumulator tiles, 16 G registers.
// Each workgroup leaves one fp32 partial in the parameter layout (Co, Ci, G); hupr_k_tmerge_wgrad_reduce sums them in a fixed order.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 tm_tr_pair(const char* base, int off0, int off1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off1));
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo;
    u.s.b = hi;
    return u.v;
}

// Wider maps (C = 128 / 256: the level-2 / level-3 merges) run through the SAME kernel: a frame of C channels is SUB = C / 64 "virtual
// frames" of 64 channels (rows of 128 bytes at a pitch of 2 C bytes — the DMA does not care), G here = real frames x SUB (8 at all
// three levels), and the Co / 64 output-channel blocks are spread over the workgroups (workgroup w: block w % NB, tile slot w / NB), each
// of which reads its 64-channel column block of dy (rows of 256 bytes at a pitch of 4 Co) and ALL of x — x is read NB times, which
// still beats the generic engine 2-3x at these sizes.  Partials are [workgroup][co 64][ci 64][virtual frame]; the reduce kernel
// scatters them to the parameter layout (Co, Ci, frames).
template <int G>
__global__ __launch_bounds__(256) void hupr_k_tmerge_wgrad_stream(const __bf16* __restrict__ x, const float* __restrict__ dy,
                                                                  float* __restrict__ part, int Bn, int HW, int SUB, int NB) {
    constexpr int C = 64, S = G + 2, SLOT = 128 * C * 2;          // stages per tile; bytes per ring slot
    __shared__ __attribute__((aligned(16))) char Ring[kTmStages][SLOT];
    __shared__ __attribute__((aligned(16))) char Db[128 * C * 2];                // bf16 dy tile [voxel][co], 16-byte chunks swizzled
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lh = lane >> 5;

    const int tiles_per_b = HW / 128, n_tiles = Bn * tiles_per_b;
    const int cob = (int)blockIdx.x % NB, slot_id = (int)blockIdx.x / NB, n_slots = (int)gridDim.x / NB;
    const int my_tiles = (slot_id < n_tiles) ? (n_tiles - 1 - slot_id) / n_slots + 1 : 0;
    const int n_stage = my_tiles * S;
    const int ct = wave >> 1, it = wave & 1;
    const int Gr = G / SUB, xpitch = SUB * C * 2, dpitch = NB * C * 4;           // real frames; bytes per voxel row of x / of dy

    f32x16 acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

    if (n_stage > 0) {
        const u32x4 xrs = {(unsigned)(unsigned long)x, (unsigned)((unsigned long)x >> 32) & 0xffffu,
                           (unsigned)((long)Bn * Gr * HW * xpitch), 0x00020000u};
        const u32x4 drs = {(unsigned)(unsigned long)dy, (unsigned)((unsigned long)dy >> 32) & 0xffffu,
                           (unsigned)((long)Bn * HW * dpitch), 0x00020000u};
        const unsigned ring_lds = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)&Ring[0][0];
        // stage n = (tile n / S, s = n % S): s < 2 -> half s of the fp32 dy tile (64 voxels x 64 co = 16 KB, four 256-byte rows per 1 KB
        // piece), else virtual frame s - 2 of the x tile (128 voxels = 16 KB, 16-byte chunks swizzled by the row on the source side as
        // in the forward kernel)
        auto dma = [&](int n) {
            const int t = slot_id + (n / S) * n_slots, s = n % S;
            const int b = t / tiles_per_b, v0 = (t % tiles_per_b) * 128;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int piece = 4 * wave + j;
                const unsigned dst = __builtin_amdgcn_readfirstlane(ring_lds + (n % kTmStages) * SLOT + piece * 1024);
                unsigned keep_;
                if (s < 2) {
                    const unsigned voff = (unsigned)(((long)b * HW + v0 + 64 * s + 4 * piece + (lane >> 4)) * dpitch) + cob * 256 + (lane & 15) * 16;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\t"
                                 "s_mov_b32 m0, %0"
                                 : "=&s"(keep_)
                                 : "s"(dst), "v"(voff), "s"(drs)
                                 : "memory");
                } else {
                    const int row = piece * 8 + (lane >> 3), f = s - 2;
                    const unsigned voff = (unsigned)((((long)b * Gr + f / SUB) * HW + v0 + row) * xpitch) + (f % SUB) * 128 +
                                          ((((lane & 7) ^ ((row >> 1) & 7)) & 7) << 4);
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\t"
                                 "s_mov_b32 m0, %0"
                                 : "=&s"(keep_)
                                 : "s"(dst), "v"(voff), "s"(xrs)
                                 : "memory");
                }
            }
        };
#define HUPR_VMCNT(N_) __builtin_amdgcn_s_waitcnt(0x0F70 | ((N_) & 15) | (((N_) >> 4) << 14))
        for (int n = 0; n < kTmStages - 1 && n < n_stage; ++n) dma(n);

        // transpose-read supplier role of this lane (wgrad_halo_bf16.hip): 16-lane group q, index sq inside it -> rows 8 lh + 4 t + (sq >> 2)
        // of a 16-voxel K step, the 8-byte segment at channel 16 (q & 1) + 4 (sq & 3) of this wave's 32-channel half
        const int q = lane >> 4, sq = lane & 15;
        int offa[2], offb[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int row = 8 * lh + 4 * tt + (sq >> 2), key = (row >> 1) & 7;       // K steps advance rows by 16: the key is unchanged
            const int ca = (32 * ct + 16 * (q & 1) + 4 * (sq & 3)) * 2, cb = (32 * it + 16 * (q & 1) + 4 * (sq & 3)) * 2;
            offa[tt] = row * 128 + ((((ca >> 4) ^ key) & 7) << 4) + (ca & 15);
            offb[tt] = row * 128 + ((((cb >> 4) ^ key) & 7) << 4) + (cb & 15);
        }

        bf16x8 a[8];
        for (int i = 0; i < my_tiles; ++i) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int n = i * S + s;
                const int younger = min(kTmStages - 2, n_stage - 1 - n) * 4;
                if (younger >= 8) HUPR_VMCNT(8); else if (younger == 4) HUPR_VMCNT(4); else HUPR_VMCNT(0);
                __syncthreads();                                  // stage n is in LDS; slot (n - 1) % 4 is free
                if (n + kTmStages - 1 < n_stage) dma(n + kTmStages - 1);
                const char* slot = &Ring[n % kTmStages][0];
                if (s < 2) {
                    // fp32 (64 voxels x 64 co) -> bf16 rows 64 s .. of Db (read by everyone from stage 2 on, behind that stage's barrier;
                    // the previous tile's last reads of Db are behind this stage's barrier)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int f = j * 256 + tid, row = 64 * s + (f >> 4), c4 = f & 15;
                        const f32x4n v = *reinterpret_cast<const f32x4n*>(slot + (long)f * 16);
                        const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                        *reinterpret_cast<bf16x4*>(Db + row * 128 + (((c4 >> 1) ^ ((row >> 1) & 7)) << 4) + ((c4 & 1) << 3)) = o;
                    }
                } else {
                    if (s == 2) {
#pragma unroll
                        for (int ks = 0; ks < 8; ++ks) a[ks] = tm_tr_pair(Db + ks * 16 * 128, offa[0], offa[1]);
                    }
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) {
                        const bf16x8 bq = tm_tr_pair(slot + ks * 16 * 128, offb[0], offb[1]);
                        acc[s - 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], bq, acc[s - 2], 0, 0, 0);
                    }
                }
            }
        }
#undef HUPR_VMCNT
    }
    // partial [workgroup][co][ci][virtual frame]: a lane holds, for its column ci and 16 rows co, the G frame values = 4 G contiguous bytes
    float* dst = part + (long)blockIdx.x * (C * C * G);
    const int ci = 32 * it + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = 32 * ct + 8 * (r >> 2) + 4 * lh + (r & 3);
        float* d = dst + ((long)co * C + ci) * G;
        if constexpr (G % 4 == 0) {
#pragma unroll
            for (int g = 0; g < G; g += 4) *reinterpret_cast<f32x4n*>(d + g) = (f32x4n){acc[g][r], acc[g + 1][r], acc[g + 2][r], acc[g + 3][r]};
        } else {
#pragma unroll
            for (int g = 0; g < G; ++g) d[g] = acc[g][r];
        }
    }
}

// 64-channel maps: the partial layout IS the parameter layout — 16 slices of the partial list per block, 16 float4 columns each
// (same summation order as the general kernel below: slice sl adds partials sl, sl + 16, ..., then the slices are added in order)
__global__ __launch_bounds__(256) void hupr_k_tmerge_wgrad_reduce4(const float* __restrict__ part, float* __restrict__ dw, int n_part,
                                                                   int n4) {
    __shared__ f32x4n sm[16][16];
    const int sl = threadIdx.x >> 4, c = threadIdx.x & 15, col = blockIdx.x * 16 + c;
    f32x4n s = {0.f, 0.f, 0.f, 0.f};
    if (col < n4)
        for (int w = sl; w < n_part; w += 16) s += reinterpret_cast<const f32x4n*>(part)[(long)w * n4 + col];
    sm[sl][c] = s;
    __syncthreads();
    if (sl == 0 && col < n4) {
        f32x4n t = sm[0][c];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += sm[k][c];
        reinterpret_cast<f32x4n*>(dw)[col] = t;
    }
}

// dw (Co, Ci, Gr) = sum over the workgroups of an output-channel block of their partials [co 64][ci 64][F], in a fixed order; element
// (co, ci, g) lives in block co / 64 at [co % 64][ci % 64][g * SUB + ci / 64].  16 slices of the workgroup list per thread block.
__global__ __launch_bounds__(256) void hupr_k_tmerge_wgrad_reduce(const float* __restrict__ part, float* __restrict__ dw, int n_slots,
                                                                  int NB, int SUB, int F, int n_out) {
    __shared__ float sm[16][16];
    const int sl = threadIdx.x >> 4, c = threadIdx.x & 15, e = blockIdx.x * 16 + c;
    float s = 0.f;
    if (e < n_out) {
        const int Gr = F / SUB, Ci = SUB * 64;
        const int g = e % Gr, ci = (e / Gr) % Ci, co = e / (Gr * Ci);
        const long src = ((long)(co & 63) * 64 + (ci & 63)) * F + g * SUB + (ci >> 6);
        const long stride = (long)NB * 64 * 64 * F;               // workgroups of one block are NB apart
        const float* p = part + (long)(co >> 6) * 64 * 64 * F + src;
        for (int w = sl; w < n_slots; w += 16) s += p[w * stride];
    }
    sm[sl][c] = s;
    __syncthreads();
    if (sl == 0 && e < n_out) {
        float t = sm[0][c];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += sm[k][c];
        dw[e] = t;
    }
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
    if (G == 8) HUPR_LAUNCH(hupr_k_tmerge_fwd_stream<8>, grid, dim3(256), 0, s, xb, wb, y, Bn, HW);
    else if (G == 4) HUPR_LAUNCH(hupr_k_tmerge_fwd_stream<4>, grid, dim3(256), 0, s, xb, wb, y, Bn, HW);
    else HUPR_LAUNCH(hupr_k_tmerge_fwd_stream<2>, grid, dim3(256), 0, s, xb, wb, y, Bn, HW);
    HUPR_LAUNCH_OK("hupr_k_tmerge_fwd_stream");
    return HUPR_OK;
}

// dy fp32 (Bn, HW, 64); wp1_bf16: the merge weight packed in mode 1 ([Ci][G - 1 - g][Co] bf16); dx bf16 (Bn, G, HW, 64).
extern "C" int hupr_tmerge_dgrad_stream_bf16(const float* dy, const void* wp1_bf16, void* dx, int Bn, int G, int HW, int Ci, int Co,
                                             hupr_stream_t stream) {
    HUPR_REQUIRE(dy && wp1_bf16 && dx && Bn > 0, "hupr_tmerge_dgrad_stream_bf16: bad argument");
    HUPR_REQUIRE(hupr_tmerge_stream_supported(G, HW, Ci, Co), "hupr_tmerge_dgrad_stream_bf16: unsupported geometry (G=%d HW=%d Ci=%d Co=%d)", G, HW, Ci, Co);
    HUPR_REQUIRE(((uintptr_t)dy & 15) == 0 && ((uintptr_t)wp1_bf16 & 15) == 0 && ((uintptr_t)dx & 15) == 0, "hupr_tmerge_dgrad_stream_bf16: misaligned pointer");
    const int tiles = Bn * (HW / 128);
    const dim3 grid((unsigned)min(tiles, 256));
    const __bf16* wb = static_cast<const __bf16*>(wp1_bf16);
    __bf16* dxb = static_cast<__bf16*>(dx);
    hipStream_t s = as_stream(stream);
    if (G == 8) HUPR_LAUNCH(hupr_k_tmerge_dgrad_stream<8>, grid, dim3(256), 0, s, dy, wb, dxb, Bn, HW);
    else if (G == 4) HUPR_LAUNCH(hupr_k_tmerge_dgrad_stream<4>, grid, dim3(256), 0, s, dy, wb, dxb, Bn, HW);
    else HUPR_LAUNCH(hupr_k_tmerge_dgrad_stream<2>, grid, dim3(256), 0, s, dy, wb, dxb, Bn, HW);
    HUPR_LAUNCH_OK("hupr_k_tmerge_dgrad_stream");
    return HUPR_OK;
}

// The streaming weight gradient also takes the wider merges: C = 64, 128 or 256 channels (Ci == Co) with G * C / 64 in {2, 4, 8}.
extern "C" int hupr_tmerge_wgrad_stream_supported(int G, int HW, int Ci, int Co) {
    if (Ci != Co || Ci % 64 != 0 || Ci > 256 || HW % 128 != 0 || G < 1) return 0;
    const int F = G * (Ci / 64);
    return (F == 8 || F == 4 || F == 2) ? 1 : 0;
}

static void tm_wgrad_grid(int Bn, int HW, int Co, int* nb, int* n_slots) {
    *nb = Co / 64;
    *n_slots = min(Bn * (HW / 128), 256 / *nb);
}

// Workspace of the streaming weight gradient: one fp32 partial [64][64][G * Ci / 64] per persistent workgroup.
extern "C" size_t hupr_tmerge_wgrad_stream_ws_bytes(int Bn, int G, int HW, int Ci, int Co) {
    if (!hupr_tmerge_wgrad_stream_supported(G, HW, Ci, Co) || Bn <= 0) return 0;
    int nb, n_slots;
    tm_wgrad_grid(Bn, HW, Co, &nb, &n_slots);
    return (size_t)nb * n_slots * 64 * 64 * (G * (Ci / 64)) * sizeof(float);
}

// x bf16 (Bn, G, HW, C), dy fp32 (Bn, HW, C) -> dw fp32 in the parameter layout (Co, Ci, G) (overwritten).  Deterministic.
extern "C" int hupr_tmerge_wgrad_stream_bf16(const void* x, const float* dy, float* dw, int Bn, int G, int HW, int Ci, int Co,
                                             void* ws, size_t ws_bytes, hupr_stream_t stream) {
    HUPR_REQUIRE(x && dy && dw && ws && Bn > 0, "hupr_tmerge_wgrad_stream_bf16: bad argument");
    HUPR_REQUIRE(hupr_tmerge_wgrad_stream_supported(G, HW, Ci, Co), "hupr_tmerge_wgrad_stream_bf16: unsupported geometry (G=%d HW=%d Ci=%d Co=%d)", G, HW, Ci, Co);
    HUPR_REQUIRE(ws_bytes >= hupr_tmerge_wgrad_stream_ws_bytes(Bn, G, HW, Ci, Co), "hupr_tmerge_wgrad_stream_bf16: workspace too small");
    HUPR_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)dw & 15) == 0 && ((uintptr_t)ws & 15) == 0, "hupr_tmerge_wgrad_stream_bf16: misaligned pointer");
    HUPR_REQUIRE((long)Bn * G * HW * Ci * 2 < 0x7fffffffL && (long)Bn * HW * Co * 4 < 0x7fffffffL, "hupr_tmerge_wgrad_stream_bf16: tensor too large for 32-bit buffer offsets");
    int nb, n_slots;
    tm_wgrad_grid(Bn, HW, Co, &nb, &n_slots);
    const int SUB = Ci / 64, F = G * SUB, n_out = Co * Ci * G;
    const __bf16* xb = static_cast<const __bf16*>(x);
    float* part = static_cast<float*>(ws);
    hipStream_t s = as_stream(stream);
    const dim3 grid((unsigned)(nb * n_slots));
    if (F == 8) HUPR_LAUNCH(hupr_k_tmerge_wgrad_stream<8>, grid, dim3(256), 0, s, xb, dy, part, Bn, HW, SUB, nb);
    else if (F == 4) HUPR_LAUNCH(hupr_k_tmerge_wgrad_stream<4>, grid, dim3(256), 0, s, xb, dy, part, Bn, HW, SUB, nb);
    else HUPR_LAUNCH(hupr_k_tmerge_wgrad_stream<2>, grid, dim3(256), 0, s, xb, dy, part, Bn, HW, SUB, nb);
    HUPR_LAUNCH_OK("hupr_k_tmerge_wgrad_stream");
    if (SUB == 1 && nb == 1 && n_out % 4 == 0)
        HUPR_LAUNCH(hupr_k_tmerge_wgrad_reduce4, dim3((n_out / 4 + 15) / 16), dim3(256), 0, s, part, dw, n_slots, n_out / 4);
    else
        HUPR_LAUNCH(hupr_k_tmerge_wgrad_reduce, dim3((n_out + 15) / 16), dim3(256), 0, s, part, dw, n_slots, nb, SUB, F, n_out);
    HUPR_LAUNCH_OK("hupr_k_tmerge_wgrad_reduce");
    return HUPR_OK;
}
