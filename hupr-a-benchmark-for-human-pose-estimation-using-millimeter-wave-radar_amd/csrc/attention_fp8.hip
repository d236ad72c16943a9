// BASELINE.json config 5: MSCSA attention forward with fp8 (OCP e4m3) MFMA operands — the measured A/B against the bf16
// flash kernel of attention_bf16.hip (scripts/attn_fp8_ab.py, profiles/r02_attn_fp8_ab.txt), for the level that carries
// 88 % of the attention flops (C = 64 channels, N = 4096 tokens).
//
// Reference semantics (models/layers.py:126-133): S[j,q] = sum_c K[j,c] Q[q,c]; P = softmax over keys j;
// out[q,c] = sum_j P[j,q] V[j,c] (+ V[q,c]).  Same keys-x-queries orientation and online softmax as the bf16 kernel.
//
//   * K, Q, V are quantised ONCE per attention with per-tensor scales s = 448 / amax (hupr_attn_quant_e4m3: one amax
//     pass + one convert pass; V is written TRANSPOSED, [channel][token], so that the P.V product finds its A operand
//     (rows = channels, K = keys) in row-major bytes);
//   * S^T = K8 Q8^T and O^T += V8^T P8^T run on v_mfma_f32_32x32x16_fp8_fp8 (the non-scaled fp8 MFMA: bf16 rate, half the
//     operand bytes in LDS and registers; the scaled 32x32x64 form is the 2x-rate one);
//   * probabilities are rounded to e4m3 as 256 p (p <= 1; three mantissa bits = 6 % per element, averaged over the keys);
//   * the fp32 accumulate, the running max / sum and the residual epilogue are unchanged.
#include "gemm_common.h"

namespace hupr {

constexpr float kLog2e8 = 1.4426950408889634f;
constexpr float kE4M3Max = 448.f, kPScale = 256.f;

// ---- per-tensor amax and quantisation ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hupr_k_absmax(const float* __restrict__ x, long n4, unsigned* __restrict__ amax_bits) {
    float m = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(amax_bits, __float_as_uint(m));      // non-negative floats order like their bits
}

__device__ __forceinline__ unsigned pack4_e4m3(float a, float b, float c, float d) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}
__device__ __forceinline__ float clamp448(float v) { return fminf(fmaxf(v, -kE4M3Max), kE4M3Max); }

// x (rows, 64) fp32 -> e4m3 bytes; TR = false: same layout; TR = true: per batch [64 channels][N tokens]
template <bool TR>
__global__ __launch_bounds__(256) void hupr_k_quant_e4m3(const float* __restrict__ x, unsigned char* __restrict__ y,
                                                         const unsigned* __restrict__ amax_bits, float* __restrict__ scale_out,
                                                         long rows, int N) {
    const float amax = __uint_as_float(*amax_bits);
    const float s = amax > 0.f ? kE4M3Max / amax : 1.f;
    if (blockIdx.x == 0 && threadIdx.x == 0) *scale_out = s;
    if constexpr (!TR) {
        const long n4 = rows * 16;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
            const float4 v = reinterpret_cast<const float4*>(x)[i];
            reinterpret_cast<unsigned*>(y)[i] = pack4_e4m3(clamp448(v.x * s), clamp448(v.y * s), clamp448(v.z * s), clamp448(v.w * s));
        }
    } else {
        // 64 tokens x 64 channels per block through LDS; writes 64-byte runs of tokens per channel
        __shared__ float t[64][65];
        const long blocks = rows / 64;
        for (long b = blockIdx.x; b < blocks; b += gridDim.x) {
            const long r0 = b * 64;                                   // first token row (batch-major rows; N % 64 == 0)
            for (int i = threadIdx.x; i < 64 * 16; i += 256) {
                const int r = i >> 4, c4 = i & 15;
                const float4 v = reinterpret_cast<const float4*>(x + (r0 + r) * 64)[c4];
                t[r][4 * c4] = v.x; t[r][4 * c4 + 1] = v.y; t[r][4 * c4 + 2] = v.z; t[r][4 * c4 + 3] = v.w;
            }
            __syncthreads();
            const long batch = r0 / N, n0 = r0 % N;
            for (int i = threadIdx.x; i < 64 * 16; i += 256) {
                const int ch = i >> 4, k4 = i & 15;
                reinterpret_cast<unsigned*>(y + (batch * 64 + ch) * (long)N + n0)[k4] =
                    pack4_e4m3(clamp448(t[4 * k4][ch] * s), clamp448(t[4 * k4 + 1][ch] * s), clamp448(t[4 * k4 + 2][ch] * s),
                               clamp448(t[4 * k4 + 3][ch] * s));
            }
            __syncthreads();
        }
    }
}

// ---- forward -----------------------------------------------------------------------------------------------------------
// LDS images: 64 rows x 64 bytes, 8-byte chunk c of row r stored at chunk c ^ (r & 7)
__device__ __forceinline__ int img8(int row, int chunk) { return row * 64 + ((chunk ^ (row & 7)) << 3); }

__global__ __launch_bounds__(256, 2) void hupr_k_attn_fwd_fp8(const unsigned char* __restrict__ K8, const unsigned char* __restrict__ Q8,
                                                              const unsigned char* __restrict__ VT8, const float* __restrict__ scales,
                                                              const float* __restrict__ Vres, float* __restrict__ out,
                                                              float* __restrict__ lse, int N) {
    constexpr int D = 64;
    __shared__ __attribute__((aligned(16))) unsigned char Ks[64 * 64];
    __shared__ __attribute__((aligned(16))) unsigned char Vt[64 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
    const long bN = (long)blockIdx.y * N;
    const int q = blockIdx.x * 128 + wave * 32 + lr;
    const float inv_kq = 1.f / (scales[0] * scales[1]);            // S = S8 / (sK sQ)
    const float c2 = kLog2e8 * inv_kq;
    long qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const long*>(Q8 + (bN + q) * D + ks * 16 + lh * 8);
    f32x16 o[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;                          // running max in RAW (scaled) score units
    // staging: thread t moves 16 bytes of row t >> 2 (two swizzled 8-byte chunks) of each image
    const int srow = tid >> 2, sc = (tid & 3) * 2;
    const unsigned char* kp = K8 + (bN + srow) * D + sc * 8;
    const unsigned char* vp = VT8 + ((long)blockIdx.y * D + srow) * N + sc * 8;
    typedef unsigned u32x4b __attribute__((ext_vector_type(4)));
    u32x4b kreg = *reinterpret_cast<const u32x4b*>(kp), vreg = *reinterpret_cast<const u32x4b*>(vp);
    for (int j0 = 0; j0 < N; j0 += 64) {
        __syncthreads();
        *reinterpret_cast<uint2*>(&Ks[img8(srow, sc)]) = make_uint2(kreg[0], kreg[1]);
        *reinterpret_cast<uint2*>(&Ks[img8(srow, sc + 1)]) = make_uint2(kreg[2], kreg[3]);
        *reinterpret_cast<uint2*>(&Vt[img8(srow, sc)]) = make_uint2(vreg[0], vreg[1]);
        *reinterpret_cast<uint2*>(&Vt[img8(srow, sc + 1)]) = make_uint2(vreg[2], vreg[3]);
        __syncthreads();
        if (j0 + 64 < N) {                                         // next tile travels while this one is multiplied
            kreg = *reinterpret_cast<const u32x4b*>(kp + (long)(j0 + 64) * D);
            vreg = *reinterpret_cast<const u32x4b*>(vp + j0 + 64);
        }
        // S^T tile (raw): rows = keys, this lane's column = its query
        f32x16 st[2];
        long ka[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ka[t][ks] = *reinterpret_cast<const long*>(&Ks[img8(32 * t + lr, ks * 2 + lh)]);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[t][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) st[t] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(ka[t][ks], qf[ks], st[t], 0, 0, 0);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[t][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
        const float nm = -m_new * c2;
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(st[t][r], c2, nm));
                st[t][r] = pv;
                sum += pv;
            }
        l_run = l_run * alpha + sum;
        m_run = m_new;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f)) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;
        }
        // O^T += V8^T P8^T.  K slot 8h + i of K-step (t, u) <-> key 32t + 16u + 8(i >> 2) + 4h + (i & 3): the probability tile in
        // its accumulator layout is the B operand; the A operand's two dwords sit at key bytes 32t + 16u + 4h and + 8 of the
        // lane's channel row of the transposed image.
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const unsigned lo = pack4_e4m3(st[t][8 * u] * kPScale, st[t][8 * u + 1] * kPScale, st[t][8 * u + 2] * kPScale, st[t][8 * u + 3] * kPScale);
                const unsigned hi = pack4_e4m3(st[t][8 * u + 4] * kPScale, st[t][8 * u + 5] * kPScale, st[t][8 * u + 6] * kPScale, st[t][8 * u + 7] * kPScale);
                const long pb = (long)(((unsigned long)hi << 32) | lo);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const int ch = 32 * ct + lr;
                    const unsigned a0 = *reinterpret_cast<const unsigned*>(&Vt[img8(ch, 4 * t + 2 * u) + 4 * lh]);
                    const unsigned a1 = *reinterpret_cast<const unsigned*>(&Vt[img8(ch, 4 * t + 2 * u + 1) + 4 * lh]);
                    const long va = (long)(((unsigned long)a1 << 32) | a0);
                    o[ct] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(va, pb, o[ct], 0, 0, 0);
                }
            }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float oscale = 1.f / (l_tot * scales[2] * kPScale);
    const long base = bN * D;
    float* dst = out + base + (long)q * D;
    const float* add = Vres ? Vres + base + (long)q * D : nullptr;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int c = 32 * ct + 8 * q4 + 4 * lh;
            float4 v = make_float4(o[ct][4 * q4] * oscale, o[ct][4 * q4 + 1] * oscale, o[ct][4 * q4 + 2] * oscale, o[ct][4 * q4 + 3] * oscale);
            if (add) {
                const float4 a = *reinterpret_cast<const float4*>(add + c);
                v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
            }
            *reinterpret_cast<float4*>(dst + c) = v;
        }
    if (lh == 0) lse[bN + q] = m_run * inv_kq + __logf(l_tot);
}

}  // namespace hupr

using namespace hupr;

extern "C" size_t hupr_attn_fp8_ws_bytes(int Bn, int N, int C) {
    // three e4m3 tensors + 3 amax words + 3 scales, 256-byte aligned pieces
    const size_t t = align_up((size_t)Bn * N * C, 256);
    return 3 * t + 256;
}

static int fp8_check(const char* who, int Bn, int N, int C, const void* ws, size_t ws_bytes) {
    HUPR_REQUIRE(ws && Bn > 0, "%s: bad argument", who);
    HUPR_REQUIRE(C == 64 && N % 128 == 0, "%s: only C = 64, N %% 128 == 0 (got C=%d N=%d)", who, C, N);
    if (ws_bytes < hupr_attn_fp8_ws_bytes(Bn, N, C)) return fail(HUPR_ERR_WORKSPACE, "%s: workspace too small", who);
    return HUPR_OK;
}

// step 1: per-tensor amax + e4m3 copies of K, Q (row-major) and V (transposed) into the workspace
extern "C" int hupr_attn_quant_fp8(const float* K, const float* Q, const float* V, int Bn, int N, int C, void* ws, size_t ws_bytes,
                                   hupr_stream_t stream) {
    if (int rc = fp8_check("hupr_attn_quant_fp8", Bn, N, C, ws, ws_bytes)) return rc;
    HUPR_REQUIRE(K && Q && V, "hupr_attn_quant_fp8: null tensor");
    hipStream_t s = as_stream(stream);
    const size_t t = align_up((size_t)Bn * N * C, 256);
    unsigned char* k8 = static_cast<unsigned char*>(ws);
    unsigned char* q8 = k8 + t;
    unsigned char* vt8 = q8 + t;
    unsigned* amax = reinterpret_cast<unsigned*>(vt8 + t);
    float* scales = reinterpret_cast<float*>(amax + 4);
    const long rows = (long)Bn * N, n4 = rows * C / 4;
    if (hipMemsetAsync(amax, 0, 16, s) != hipSuccess) return fail(HUPR_ERR_LAUNCH, "hupr_attn_quant_fp8: memset failed");
    const dim3 rg((unsigned)min((long)2048, (n4 + 255) / 256));
    HUPR_LAUNCH(hupr_k_absmax, rg, dim3(256), 0, s, K, n4, amax);
    HUPR_LAUNCH(hupr_k_absmax, rg, dim3(256), 0, s, Q, n4, amax + 1);
    HUPR_LAUNCH(hupr_k_absmax, rg, dim3(256), 0, s, V, n4, amax + 2);
    HUPR_LAUNCH(hupr_k_quant_e4m3<false>, rg, dim3(256), 0, s, K, k8, amax, scales, rows, N);
    HUPR_LAUNCH(hupr_k_quant_e4m3<false>, rg, dim3(256), 0, s, Q, q8, amax + 1, scales + 1, rows, N);
    HUPR_LAUNCH(hupr_k_quant_e4m3<true>, dim3((unsigned)min((long)2048, rows / 64)), dim3(256), 0, s, V, vt8, amax + 2, scales + 2, rows, N);
    HUPR_LAUNCH_OK("hupr_k_quant_e4m3");
    return HUPR_OK;
}

// step 2: the attention itself on the quantised copies; Vres = fp32 V for the residual form (exact add) or null
extern "C" int hupr_attn_fwd_fp8_quantized(const void* ws, const float* Vres, float* out, float* lse, int Bn, int N, int C, size_t ws_bytes,
                                           hupr_stream_t stream) {
    if (int rc = fp8_check("hupr_attn_fwd_fp8_quantized", Bn, N, C, ws, ws_bytes)) return rc;
    HUPR_REQUIRE(out && lse, "hupr_attn_fwd_fp8_quantized: null output");
    const size_t t = align_up((size_t)Bn * N * C, 256);
    const unsigned char* k8 = static_cast<const unsigned char*>(ws);
    const unsigned char* q8 = k8 + t;
    const unsigned char* vt8 = q8 + t;
    const float* scales = reinterpret_cast<const float*>(vt8 + t + 16);
    HUPR_LAUNCH(hupr_k_attn_fwd_fp8, dim3(N / 128, Bn), dim3(256), 0, as_stream(stream), k8, q8, vt8, scales, Vres, out, lse, N);
    HUPR_LAUNCH_OK("hupr_k_attn_fwd_fp8");
    return HUPR_OK;
}

extern "C" int hupr_attn_fwd_fp8(const float* K, const float* Q, const float* V, int residual, float* out, float* lse, int Bn, int N,
                                 int C, void* ws, size_t ws_bytes, hupr_stream_t stream) {
    if (int rc = hupr_attn_quant_fp8(K, Q, V, Bn, N, C, ws, ws_bytes, stream)) return rc;
    return hupr_attn_fwd_fp8_quantized(ws, residual ? V : nullptr, out, lse, Bn, N, C, ws_bytes, stream);
}
