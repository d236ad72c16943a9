// PRGCN feature products on the fp32 matrix pipe (reference models/gcn_networks.py:23-29: support = W . x per sample, F = 1024
// features x 14 key-points stored 16 wide), forward, input gradient and weight gradient.
//
// Why not the generic fp32 engine (gemm_f32.hip): as a batched GEMM the forward is 32 problems of M 1024 x N 16 x K 1024 — a
// 128 x 32 tile per workgroup walks K = 1024 alone, half of every 32-wide tile is padding, 0.36 of the fp32 MFMA rate (43 us
// for 1 GFLOP at B = 32).  Here the batch folds into the MFMA's N axis where the data lies: a 32-column B operand is TWO samples'
// sixteen key-point slots, read in place from (B, F, 16) — no transposed copy — and a workgroup owns 64 features x 2 samples with
// the reduction axis split over its two wave pairs (256 workgroups at B = 32, one round of the chip).
//   hupr_k_gcn_wx<false>   t[b][f][n]  = sum_g W[f][g] x[b][g][n]
//   hupr_k_gcn_wx<true>    dx[b][g][n] = sum_f W[f][g] dt[b][f][n]
//   hupr_k_gcn_dw          dW[f][g]    = sum_{b, n} dt[b][f][n] x[b][g][n]      (the batch folds into the reduction axis)
// v_mfma_f32_32x32x2_f32 rounds like an fmaf chain; the order of the reduction index inside a wave is a fixed permutation
// (eight-element groups: lane half h takes elements 4 h .. 4 h + 3), the two K halves are added once at the end: deterministic.
#include "gemm_common.h"

namespace hupr {

typedef float f32x4g __attribute__((ext_vector_type(4)));

// lane (c = lane & 31, h = lane >> 5) holds D[row 8 (r >> 2) + 4 h + (r & 3)][column c], r = 0..15
template <bool TRANS>
__global__ __launch_bounds__(256) void hupr_k_gcn_wx(const float* __restrict__ Wm, const float* __restrict__ x, float* __restrict__ t,
                                                      int Bn, int F) {
    __shared__ float red[2][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mi = wave & 1, kh = wave >> 1, c = lane & 31, h = lane >> 5;
    const int m = blockIdx.x * 64 + 32 * mi + c;                       // A row of this lane
    const int b = min(2 * (int)blockIdx.y + (c >> 4), Bn - 1);         // B column of this lane: (sample, key-point slot)
    const float* xb = x + (long)b * F * 16 + (c & 15);
    const int kbeg = kh * (F >> 1), kend = kbeg + (F >> 1);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // chunks of 32 reduction elements (16 MFMAs = 1 024 matrix-pipe cycles), the operands of chunk i + 1 in flight under chunk i
    float a[2][16], bv[2][16];
#define HUPR_GCN_LOAD(SET_, KK_)                                                                          \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                       \
        const int k0 = (KK_) + 8 * u + 4 * h;                                                             \
        if constexpr (!TRANS) {                                                                           \
            const f32x4g av = *reinterpret_cast<const f32x4g*>(Wm + (long)m * F + k0);                    \
            a[SET_][4 * u] = av[0]; a[SET_][4 * u + 1] = av[1]; a[SET_][4 * u + 2] = av[2]; a[SET_][4 * u + 3] = av[3]; \
        } else {                                                                                          \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) a[SET_][4 * u + j] = Wm[(long)(k0 + j) * F + m]; \
        }                                                                                                 \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) bv[SET_][4 * u + j] = xb[(long)(k0 + j) * 16];      \
    }
#define HUPR_GCN_MMA(SET_)                                                                                \
    _Pragma("unroll") for (int j = 0; j < 16; ++j)                                                        \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[SET_][j], bv[SET_][j], acc, 0, 0, 0);
    HUPR_GCN_LOAD(0, kbeg)
    for (int kk = kbeg; kk < kend; kk += 64) {           // F / 2 is a multiple of 32
        if (kk + 32 < kend) { HUPR_GCN_LOAD(1, kk + 32) }
        HUPR_GCN_MMA(0)
        if (kk + 32 < kend) {
            if (kk + 64 < kend) { HUPR_GCN_LOAD(0, kk + 64) }
            HUPR_GCN_MMA(1)
        }
    }
#undef HUPR_GCN_LOAD
#undef HUPR_GCN_MMA
    if (kh == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[mi][r][lane] = acc[r];
    }
    __syncthreads();
    if (kh == 0 && 2 * (int)blockIdx.y + (c >> 4) < Bn) {
        float* tb = t + (long)b * F * 16 + (c & 15);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = blockIdx.x * 64 + 32 * mi + 8 * (r >> 2) + 4 * h + (r & 3);
            tb[(long)row * 16] = acc[r] + red[mi][r][lane];
        }
    }
}

// one 32 x 32 tile of dW per wave, 64 x 64 per workgroup; reduction index (b, n) in eight-element groups as above
__global__ __launch_bounds__(256) void hupr_k_gcn_dw(const float* __restrict__ dt, const float* __restrict__ x, float* __restrict__ dW,
                                                      int Bn, int F) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int f = blockIdx.y * 64 + 32 * (wave >> 1) + c;             // A row (as an A operand lane)
    const int g = blockIdx.x * 64 + 32 * (wave & 1) + c;              // B column
    const float* ap = dt + (long)f * 16 + 4 * h;
    const float* bp = x + (long)g * 16 + 4 * h;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // two samples (32 reduction elements, 16 MFMAs) per step, the next pair's operands in flight under the current one
    f32x4g av[2][4], bw[2][4];
#define HUPR_GCN_LOAD(SET_, B_)                                                                           \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                       \
        const long o = (long)min((B_) + (u >> 1), Bn - 1) * F * 16 + 8 * (u & 1);                         \
        av[SET_][u] = *reinterpret_cast<const f32x4g*>(ap + o);                                           \
        bw[SET_][u] = *reinterpret_cast<const f32x4g*>(bp + o);                                           \
        if ((B_) + (u >> 1) >= Bn) av[SET_][u] = (f32x4g){0.f, 0.f, 0.f, 0.f};      /* odd batch: the last pair's second sample */ \
    }
#define HUPR_GCN_MMA(SET_)                                                                                \
    _Pragma("unroll") for (int u = 0; u < 4; ++u)                                                         \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                     \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[SET_][u][i], bw[SET_][u][i], acc, 0, 0, 0);
    HUPR_GCN_LOAD(0, 0)
    for (int b = 0; b < Bn; b += 4) {
        if (b + 2 < Bn) { HUPR_GCN_LOAD(1, b + 2) }
        HUPR_GCN_MMA(0)
        if (b + 2 < Bn) {
            if (b + 4 < Bn) { HUPR_GCN_LOAD(0, b + 4) }
            HUPR_GCN_MMA(1)
        }
    }
#undef HUPR_GCN_LOAD
#undef HUPR_GCN_MMA
    float* out = dW + (long)(blockIdx.y * 64 + 32 * (wave >> 1)) * F + g;
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(long)(8 * (r >> 2) + 4 * h + (r & 3)) * F] = acc[r];
}

}  // namespace hupr

using namespace hupr;

extern "C" int hupr_gcn_wx_f32(const float* W, const float* x, float* t, int Bn, int F, int ld, int trans_w, hupr_stream_t stream) {
    HUPR_REQUIRE(W && x && t && Bn > 0, "hupr_gcn_wx_f32: bad argument");
    HUPR_REQUIRE(ld == 16 && F > 0 && F % 64 == 0 /* F / 2 % 32 == 0 */, "hupr_gcn_wx_f32: needs ld = 16 and F %% 64 == 0 (got ld=%d F=%d)", ld, F);
    const dim3 grid(F / 64, (Bn + 1) / 2);
    if (trans_w) HUPR_LAUNCH(hupr_k_gcn_wx<true>, grid, dim3(256), 0, as_stream(stream), W, x, t, Bn, F);
    else HUPR_LAUNCH(hupr_k_gcn_wx<false>, grid, dim3(256), 0, as_stream(stream), W, x, t, Bn, F);
    HUPR_LAUNCH_OK("hupr_k_gcn_wx");
    return HUPR_OK;
}

extern "C" int hupr_gcn_dw_f32(const float* dt, const float* x, float* dW, int Bn, int F, int ld, hupr_stream_t stream) {
    HUPR_REQUIRE(dt && x && dW && Bn > 0, "hupr_gcn_dw_f32: bad argument");
    HUPR_REQUIRE(ld == 16 && F > 0 && F % 64 == 0, "hupr_gcn_dw_f32: needs ld = 16 and F %% 64 == 0 (got ld=%d F=%d)", ld, F);
    HUPR_LAUNCH(hupr_k_gcn_dw, dim3(F / 64, F / 64), dim3(256), 0, as_stream(stream), dt, x, dW, Bn, F);
    HUPR_LAUNCH_OK("hupr_k_gcn_dw");
    return HUPR_OK;
}
