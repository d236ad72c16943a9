// The 1 x 1 projections of an MSCSA level as a streaming kernel (round 6).
//
// Reference models/layers.py:150-163: each of the two maps of a level (B, C, H, W) goes through four bias-free 1 x 1 convolutions
// (phi / theta, cross / self) whose outputs are the keys and queries of the level's four attentions.  functional.MSCSALevelFn runs the
// four of a map as ONE product  Y[M][4C] = X[M][C] . Wc^T  (M = B H W rows, Wc the concatenated (4C, C) matrix, query rows pre-scaled
// by log2 e) and stores Y as the bf16 operands the attention kernels read.  Through the generic implicit-GEMM engine that product
// ran at 2.2 TB/s (level 1: 57 us for 100 MB; K = C = 64 is two K-steps per tile, the engine's fp32 -> bf16 LDS staging dominates).
// It is a pure stream: 2 C flop per byte moved.  Here:
//   * a workgroup keeps a 256-row block of Wc (256 output channels x C, bf16, 16-byte chunks XOR-swizzled) in LDS for its whole life;
//   * a wave owns 16 rows of X at a time: a lane (row = lane & 15, kq = lane >> 4) loads the eight fp32 channels 32 ks + 8 kq .. + 7 of
//     its row straight from global memory (two 16-byte loads per K-step, whole 4 C-byte rows per wave), rounds them to bf16 — the B
//     operand of v_mfma_f32_16x16x32_bf16 — with the NEXT 16 rows already in flight;
//   * D'[channel][row] = Wc X^T: a lane ends up with four consecutive output channels of its row; a v_permlane16_swap per block pair
//     turns them into 16-byte stores, 64 contiguous bytes per row and instruction.
// No LDS traffic for X, no barriers in the loop, registers small enough for 5+ waves per SIMD.
// Same operand roundings as the engine (X and Wc to bf16, fp32 accumulate, one rounding of Y); the fp32 summation order inside a row
// differs (two 32-channel K-steps per MFMA chain instead of the engine's 16-channel steps): equal up to one bf16 step on a few outputs.
#include "gemm_common.h"

namespace hupr {

typedef __bf16 pj_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 pj_bf16x4 __attribute__((ext_vector_type(4)));
typedef float pj_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 pj_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned pj_u32x4 __attribute__((ext_vector_type(4)));

template <int C>
__device__ __forceinline__ int pj_key(int row) { return C == 64 ? ((row >> 1) & 7) : (row & 15); }

// NT threads per workgroup: the level-1 map is only 512 rows per CU (128 KB in, 256 KB out) — the weight block (64 KB of fp32 source per
// workgroup) is fetched ONCE per CU by one 1024-thread workgroup instead of once per 256-thread workgroup (the first build: 1 024
// workgroups, 64 MB of weight reads beside 100 MB of data, 34.8 us).
template <int C, int NT>
__global__ __launch_bounds__(NT) void hupr_k_mscsa_proj_fwd(const float* __restrict__ X, const float* __restrict__ Wc,
                                                            __bf16* __restrict__ Y, long M, int n_row_groups) {
    constexpr int KS = C / 32, CH = C / 8, NB = 256, NW = NT / 64;      // K-steps per row, 16-byte chunks per weight row, output channels and waves per workgroup
    __shared__ __attribute__((aligned(16))) __bf16 Ws[NB * C];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, idx = lane & 15, kq = lane >> 4;
    const int cblk = blockIdx.x / n_row_groups, rg = blockIdx.x - cblk * n_row_groups;      // output column block (256 channels), row group
    const int col0 = cblk * NB, ldy = 4 * C;
    // this workgroup's 256 rows of Wc -> LDS as bf16: consecutive threads convert consecutive 32-byte pieces (coalesced)
    {
        pj_f32x4 wa[NB * CH / NT], wb[NB * CH / NT];
#pragma unroll
        for (int i = 0; i < NB * CH / NT; ++i) {
            const int q = tid + i * NT, row = q / CH, c = q % CH;
            const float* wr = Wc + (long)(col0 + row) * C + 8 * c;
            wa[i] = *reinterpret_cast<const pj_f32x4*>(wr);
            wb[i] = *reinterpret_cast<const pj_f32x4*>(wr + 4);
        }
#pragma unroll
        for (int i = 0; i < NB * CH / NT; ++i) {
            const int q = tid + i * NT, row = q / CH, c = q % CH;
            const pj_bf16x8 v = {(__bf16)wa[i][0], (__bf16)wa[i][1], (__bf16)wa[i][2], (__bf16)wa[i][3],
                                 (__bf16)wb[i][0], (__bf16)wb[i][1], (__bf16)wb[i][2], (__bf16)wb[i][3]};
            *reinterpret_cast<pj_bf16x8*>(&Ws[row * C + ((c ^ pj_key<C>(row)) << 3)]) = v;
        }
    }
    // A-operand addresses: output channel 16 cb + idx, chunk 4 ks + kq
    int woff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) woff[ks] = idx * C + (((4 * ks + kq) ^ pj_key<C>(idx)) << 3);      // + 16 cb * C (the key of row 16 cb + idx is idx's)
    const long stride = (long)n_row_groups * (NW * 16);
    long m0 = (long)rg * (NW * 16) + wave * 16;
    pj_f32x4 raw[KS][2];
    auto issue = [&](long m) {
        const bool ok = m + idx < M;
        const float* xr = X + (ok ? (m + idx) : 0) * C + 8 * kq;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            raw[ks][0] = *reinterpret_cast<const pj_f32x4*>(xr + 32 * ks);
            raw[ks][1] = *reinterpret_cast<const pj_f32x4*>(xr + 32 * ks + 4);
        }
    };
    if (m0 < M) issue(m0);                                       // (the first rows travel beside the weight block)
    __syncthreads();
    for (; m0 < M; m0 += stride) {
        asm volatile("" ::: "memory");      // the weight fragments are re-read from LDS per row group (hoisted out of the loop they would
                                            // occupy 16 KS x 4 registers: 128 / 256 / 512 at C = 64 / 128 / 256, i.e. one or two waves per SIMD)
        pj_bf16x8 xf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            xf[ks] = (pj_bf16x8){(__bf16)raw[ks][0][0], (__bf16)raw[ks][0][1], (__bf16)raw[ks][0][2], (__bf16)raw[ks][0][3],
                                 (__bf16)raw[ks][1][0], (__bf16)raw[ks][1][1], (__bf16)raw[ks][1][2], (__bf16)raw[ks][1][3]};
        if (m0 + stride < M) issue(m0 + stride);                 // the next 16 rows travel while these are multiplied
        const bool ok = m0 + idx < M;
        // v_permlane16_swap trades the odd 16-lane rows of block cb's dwords for the even rows of block cb + 1's: a kq-even lane then
        // holds channels 4 kq .. 4 kq + 7 of block cb, a kq-odd lane channels 4 (kq - 1) .. + 7 of block cb + 1 — ONE 16-byte store per
        // lane and block pair, 64 contiguous bytes per row and instruction (8-byte stores: 32, and 30.2 us at level 1)
        __bf16* yr = Y + (m0 + idx) * ldy + col0 + ((kq & 1) ? 16 + 4 * (kq - 1) : 4 * kq);
#pragma unroll
        for (int cb0 = 0; cb0 < 16; cb0 += 4) {                   // four independent accumulator chains at a time
            pj_f32x4 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = (pj_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const pj_bf16x8 wf = *reinterpret_cast<const pj_bf16x8*>(&Ws[(cb0 + i) * 16 * C + woff[ks]]);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[ks], acc[i], 0, 0, 0);
                }
            unsigned pk[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const pj_bf16x2 lo = {(__bf16)acc[i][0], (__bf16)acc[i][1]}, hi = {(__bf16)acc[i][2], (__bf16)acc[i][3]};
                pk[i][0] = __builtin_bit_cast(unsigned, lo);
                pk[i][1] = __builtin_bit_cast(unsigned, hi);
            }
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
                const auto r0 = __builtin_amdgcn_permlane16_swap(pk[i][0], pk[i + 1][0], false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(pk[i][1], pk[i + 1][1], false, false);
                if (ok) *reinterpret_cast<pj_u32x4*>(yr + 16 * (cb0 + i)) = (pj_u32x4){r0[0], r1[0], r0[1], r1[1]};
            }
        }
    }
}

// Input gradient of the same product (functional.MSCSALevelFn.backward): dX[M][C] = dY[M][4C] . Wc (+ dV[M][C]), fp32 in and out, operands
// rounded to bf16 like the engine's (dY is the fp32 gradient tensor the attention backward kernels wrote into: dK / dQ column blocks).
// Through the engine: 61.6 us at level 1 for 201 MB (3.3 TB/s).  Same stream shape as the forward with the roles of C and 4C swapped:
// Wc^T (C rows x 4C) lives in LDS as bf16, a lane loads the eight fp32 gradients 32 ks + 8 kq .. + 7 of its row per K-step — K is walked
// in chunks of 256 (eight K-steps, 64 load registers) with the next chunk (of this or the next 16 rows) in flight —, and ends up with
// four consecutive fp32 outputs per channel block: 16-byte loads of the residual and 16-byte stores, 64 contiguous bytes per row.
template <int C, int NT>
__global__ __launch_bounds__(NT) void hupr_k_mscsa_proj_dgrad(const float* __restrict__ dY, const float* __restrict__ Wc,
                                                              const float* __restrict__ res, float* __restrict__ dX, long M,
                                                              int n_row_groups) {
    constexpr int K4 = 4 * C, CHK = 8, NCHUNK = K4 / (32 * CHK), NCB = C / 16, NW = NT / 64, CHW = K4 / 8;      // CHW: 16-byte chunks per LDS row
    __shared__ __attribute__((aligned(16))) __bf16 Wt[C * K4];                                                     // [output channel n][k], chunk-swizzled by n & 15
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, idx = lane & 15, kq = lane >> 4;
    // Wc (4C, C) fp32 -> Wt (C, 4C) bf16: thread reads four consecutive n of one k (coalesced), writes four 2-byte elements
    {
        constexpr int NQ = K4 * (C / 4) / NT;                     // 8 loads per thread, all in flight
        pj_f32x4 w[NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = tid + i * NT, k = q / (C / 4), n4 = (q - k * (C / 4)) * 4;
            w[i] = *reinterpret_cast<const pj_f32x4*>(Wc + (long)k * C + n4);
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = tid + i * NT, k = q / (C / 4), n4 = (q - k * (C / 4)) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n4 + j;
                Wt[n * K4 + ((((k >> 3) ^ (n & 15)) & (CHW - 1)) << 3) + (k & 7)] = (__bf16)w[i][j];
            }
        }
    }
    const long stride = (long)n_row_groups * (NW * 16);
    long m0 = (long)blockIdx.x * (NW * 16) + wave * 16;
    pj_f32x4 raw[CHK][2];
    auto issue = [&](long m, int chunk) {
        const bool ok = m + idx < M;
        const float* xr = dY + (ok ? (m + idx) : 0) * K4 + chunk * (32 * CHK) + 8 * kq;
#pragma unroll
        for (int ks = 0; ks < CHK; ++ks) {
            raw[ks][0] = *reinterpret_cast<const pj_f32x4*>(xr + 32 * ks);
            raw[ks][1] = *reinterpret_cast<const pj_f32x4*>(xr + 32 * ks + 4);
        }
    };
    if (m0 < M) issue(m0, 0);
    __syncthreads();
    for (; m0 < M; m0 += stride) {
        pj_f32x4 acc[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[cb] = (pj_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int chunk = 0; chunk < NCHUNK; ++chunk) {
            asm volatile("" ::: "memory");                    // (weight fragments re-read from LDS, not hoisted into registers)
            pj_bf16x8 xf[CHK];
#pragma unroll
            for (int ks = 0; ks < CHK; ++ks)
                xf[ks] = (pj_bf16x8){(__bf16)raw[ks][0][0], (__bf16)raw[ks][0][1], (__bf16)raw[ks][0][2], (__bf16)raw[ks][0][3],
                                     (__bf16)raw[ks][1][0], (__bf16)raw[ks][1][1], (__bf16)raw[ks][1][2], (__bf16)raw[ks][1][3]};
            if (chunk + 1 < NCHUNK) issue(m0, chunk + 1);      // the next 256 gradients of these rows, or the first of the next rows
            else if (m0 + stride < M) issue(m0 + stride, 0);
#pragma unroll
            for (int ks = 0; ks < CHK; ++ks)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    const int c16 = 4 * (chunk * CHK + ks) + kq;                          // 16-byte chunk of row n = 16 cb + idx (key n & 15 = idx)
                    const pj_bf16x8 wf = *reinterpret_cast<const pj_bf16x8*>(&Wt[(16 * cb + idx) * K4 + ((c16 ^ idx) << 3)]);
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[ks], acc[cb], 0, 0, 0);
                }
        }
        if (m0 + idx < M) {
            const long o = (m0 + idx) * C + 4 * kq;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                pj_f32x4 v = acc[cb];
                if (res) v += *reinterpret_cast<const pj_f32x4*>(res + o + 16 * cb);
                *reinterpret_cast<pj_f32x4*>(dX + o + 16 * cb) = v;
            }
        }
    }
}

}  // namespace hupr

using namespace hupr;

// 1 where hupr_mscsa_proj_fwd_bf16 applies: C in {64, 128} (levels 1 and 2; the 8 192 rows of level 3 stay on the engine), at least
// one 16-row group
extern "C" int hupr_mscsa_proj_supported(long M, int C) { return (M >= 16 && (C == 64 || C == 128)) ? 1 : 0; }

// Y (M, 4 C) bf16 = X (M, C) fp32 . Wc^T, Wc (4 C, C) fp32: the four 1 x 1 projections of one map of an MSCSA level
// (reference models/layers.py:150-157) as one product; rows = the B H W pixels of the map, channels-last.
extern "C" int hupr_mscsa_proj_fwd_bf16(const float* X, const float* Wc, void* Y, long M, int C, hupr_stream_t stream) {
    HUPR_REQUIRE(X && Wc && Y, "hupr_mscsa_proj_fwd_bf16: null pointer");
    HUPR_REQUIRE(hupr_mscsa_proj_supported(M, C), "hupr_mscsa_proj_fwd_bf16: unsupported shape M=%ld C=%d", M, C);
    HUPR_REQUIRE(((uintptr_t)X & 15) == 0 && ((uintptr_t)Wc & 15) == 0 && ((uintptr_t)Y & 15) == 0, "hupr_mscsa_proj_fwd_bf16: misaligned buffer");
    const int nb = 4 * C / 256;                                   // column blocks of 256 output channels
    hipStream_t s = as_stream(stream);
    __bf16* y = static_cast<__bf16*>(Y);
    if (C == 64) {                                                // one 1024-thread workgroup per CU (16 waves x 16 rows per step)
        const long groups = (M + 255) / 256;
        const int nrg = (int)(groups < 256 ? groups : 256);
        HUPR_LAUNCH((hupr_k_mscsa_proj_fwd<64, 1024>), dim3((unsigned)(nb * nrg)), dim3(1024), 0, s, X, Wc, y, M, nrg);
    } else {                                                      // two column blocks: two 512-thread workgroups per CU
        const long groups = (M + 127) / 128;
        const int nrg = (int)(groups < 256 ? groups : 256);
        HUPR_LAUNCH((hupr_k_mscsa_proj_fwd<128, 512>), dim3((unsigned)(nb * nrg)), dim3(512), 0, s, X, Wc, y, M, nrg);
    }
    HUPR_LAUNCH_OK("hupr_k_mscsa_proj_fwd");
    return HUPR_OK;
}

// level 1 only (C = 64): at C = 128 the transposed 128 KB weight block costs more than the engine's whole launch (34.6 vs 20.7 us measured)
extern "C" int hupr_mscsa_proj_dgrad_supported(long M, int C) { return (M >= 16 && C == 64) ? 1 : 0; }

// dX (M, C) fp32 = dY (M, 4 C) fp32 . Wc (+ res (M, C) fp32 or null): the input gradient of the four projections of one map, the value
// gradient dV of the map's attentions as its residual term (functional.MSCSALevelFn.backward).  res may alias nothing it writes.
extern "C" int hupr_mscsa_proj_dgrad_f32(const float* dY, const float* Wc, const float* res, float* dX, long M, int C, hupr_stream_t stream) {
    HUPR_REQUIRE(dY && Wc && dX, "hupr_mscsa_proj_dgrad_f32: null pointer");
    HUPR_REQUIRE(hupr_mscsa_proj_dgrad_supported(M, C), "hupr_mscsa_proj_dgrad_f32: unsupported shape M=%ld C=%d", M, C);
    HUPR_REQUIRE(((uintptr_t)dY & 15) == 0 && ((uintptr_t)Wc & 15) == 0 && ((uintptr_t)dX & 15) == 0 && ((uintptr_t)res & 15) == 0,
                 "hupr_mscsa_proj_dgrad_f32: misaligned buffer");
    hipStream_t s = as_stream(stream);
    const long groups = (M + 127) / 128;                          // 512-thread workgroups: 8 waves x 16 rows per step, one per CU
    const int nrg = (int)(groups < 256 ? groups : 256);
    HUPR_LAUNCH((hupr_k_mscsa_proj_dgrad<64, 512>), dim3((unsigned)nrg), dim3(512), 0, s, dY, Wc, res, dX, M, nrg);
    HUPR_LAUNCH_OK("hupr_k_mscsa_proj_dgrad");
    return HUPR_OK;
}
