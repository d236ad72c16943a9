// BatchNorm (train/eval, fwd/bwd), scale-shift-activation, PReLU — all on channels-last
// fp32 activations x[row][C] (row = (b,d,h,w) voxel).  HBM-bound elementwise/reduction kernels:
// float4 accesses, per-thread partial sums over short row runs, cross-thread combination in
// double (BatchNorm variance is formed from E[x^2]-E[x]^2, so the sums themselves must be tight).
//
// Mirrors the semantics of nn.BatchNorm3d / nn.ReLU / nn.PReLU as used by the reference's
// BasicBlock3D (models/layers.py:40-70) and BasicBlock2D (models/layers.py:8-38).
#include "hupr_common.h"

namespace hupr {

constexpr int kStatBlocks = 256;

// ------------------------------------------------------------------------------------------
// column statistics: for every channel c: S1 = sum_r f(r,c), S2 = sum_r g(r,c)
//   MODE 0 (forward) : f = x,             g = x*x
//   MODE 1 (backward): f = dy',           g = dy' * xhat      dy' = dy * [y > 0] if y given
// partial[blk][2][C] doubles
// ------------------------------------------------------------------------------------------
template <int MODE, typename T>
__global__ __launch_bounds__(256) void hupr_k_colstats(const T* __restrict__ x, const T* __restrict__ dy,
                                                       const T* __restrict__ y,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, long M, int C,
                                                       double* __restrict__ partial) {
    extern __shared__ double sh[];   // [2][C]
    const int tid = threadIdx.x;
    const int c4n = C >> 2;                      // float4 per row
    const int rows_per_pass = 256 / c4n;         // C <= 1024
    const int c4 = tid % c4n, rsub = tid / c4n;
    for (int i = tid; i < 2 * C; i += 256) sh[i] = 0.0;
    __syncthreads();
    const long rows_per_block = (M + gridDim.x - 1) / gridDim.x;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(M, r0 + rows_per_block);
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    float mu[4] = {0, 0, 0, 0}, is[4] = {1, 1, 1, 1};
    if (MODE == 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { mu[k] = mean[c4 * 4 + k]; is[k] = invstd[c4 * 4 + k]; }
    }
    if (rsub < rows_per_pass) {
        for (long r = r0 + rsub; r < r1; r += rows_per_pass) {
            const float4 xv = ld_act4(x + r * C + c4 * 4);
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { s1[k] += xs[k]; s2[k] = fmaf(xs[k], xs[k], s2[k]); }
            } else {
                const float4 gv = ld_act4(dy + r * C + c4 * 4);
                float gs[4] = {gv.x, gv.y, gv.z, gv.w};
                if (y) {
                    const float4 yv = ld_act4(y + r * C + c4 * 4);
                    const float ys[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) gs[k] = ys[k] > 0.f ? gs[k] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    s1[k] += gs[k];
                    s2[k] = fmaf(gs[k], (xs[k] - mu[k]) * is[k], s2[k]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            atomicAdd(&sh[c4 * 4 + k], (double)s1[k]);
            atomicAdd(&sh[C + c4 * 4 + k], (double)s2[k]);
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * C; i += 256) partial[(long)blockIdx.x * 2 * C + i] = sh[i];
}


// sum partial[b][2][C] over b for channel c; 256 threads = 4 row-groups x 64 channels per block
__device__ __forceinline__ bool reduce_partials(const double* __restrict__ partial, int nblk, int C, int& c,
                                                double& s1, double& s2) {
    __shared__ double sh1[4][64], sh2[4][64];
    const int g = threadIdx.x >> 6, cl = threadIdx.x & 63;
    c = blockIdx.x * 64 + cl;
    double a = 0.0, b2 = 0.0;
    if (c < C) {
        for (int b = g; b < nblk; b += 4) {
            a += partial[(long)b * 2 * C + c];
            b2 += partial[(long)b * 2 * C + C + c];
        }
    }
    sh1[g][cl] = a;
    sh2[g][cl] = b2;
    __syncthreads();
    s1 = (sh1[0][cl] + sh1[1][cl]) + (sh1[2][cl] + sh1[3][cl]);
    s2 = (sh2[0][cl] + sh2[1][cl]) + (sh2[2][cl] + sh2[3][cl]);
    return g == 0 && c < C;
}

// forward finalize: batch mean / biased var -> save_mean, save_invstd, scale, shift; running stats
__global__ void hupr_k_bn_finalize_fwd(const double* __restrict__ partial, int nblk, long M, int C,
                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                       float* __restrict__ running_mean, float* __restrict__ running_var,
                                       float momentum, float eps, float* __restrict__ save_mean,
                                       float* __restrict__ save_invstd, float* __restrict__ scale,
                                       float* __restrict__ shift) {
    int c;
    double s1, s2;
    if (!reduce_partials(partial, nblk, C, c, s1, s2)) return;
    const double mean = s1 / (double)M;
    double var = s2 / (double)M - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    save_mean[c] = (float)mean;
    save_invstd[c] = invstd;
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;
    if (running_mean) {
        const double unbiased = (M > 1) ? var * (double)M / (double)(M - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

__global__ void hupr_k_bn_eval_params(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                      int C, float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = 1.0f / sqrtf(rv[c] + eps);
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
}

// y = act(x1*s1 + t1 (+ x2*s2 + t2)) ; act: 0 none, 1 relu
template <typename T>
__global__ __launch_bounds__(256) void hupr_k_scale_shift_act(const T* __restrict__ x1,
                                                              const float* __restrict__ s1,
                                                              const float* __restrict__ t1,
                                                              const T* __restrict__ x2,
                                                              const float* __restrict__ s2,
                                                              const float* __restrict__ t2,
                                                              T* __restrict__ y, long n4, int C, int act) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int c = (int)((i * 4) % C);
        float4 v = ld_act4(x1 + i * 4);
        const float4 a = *reinterpret_cast<const float4*>(s1 + c), b = *reinterpret_cast<const float4*>(t1 + c);
        v.x = fmaf(v.x, a.x, b.x); v.y = fmaf(v.y, a.y, b.y);
        v.z = fmaf(v.z, a.z, b.z); v.w = fmaf(v.w, a.w, b.w);
        if (x2) {
            const float4 u = ld_act4(x2 + i * 4);
            const float4 a2 = *reinterpret_cast<const float4*>(s2 + c), b2 = *reinterpret_cast<const float4*>(t2 + c);
            v.x += fmaf(u.x, a2.x, b2.x); v.y += fmaf(u.y, a2.y, b2.y);
            v.z += fmaf(u.z, a2.z, b2.z); v.w += fmaf(u.w, a2.w, b2.w);
        }
        if (act == 1) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        st_act4(y + i * 4, v);
    }
}

// backward finalize + apply:
//   dgamma = S2, dbeta = S1,  dx = gamma*invstd*(dy' - S1/M - xhat*S2/M)       (train)
//   dx = gamma*invstd*dy'                                                      (eval)
__global__ void hupr_k_bn_finalize_bwd(const double* __restrict__ partial, int nblk, int C,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ sums /* [2][C] floats */) {
    int c;
    double s1, s2;
    if (!reduce_partials(partial, nblk, C, c, s1, s2)) return;
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
    sums[c] = (float)s1;
    sums[C + c] = (float)s2;
}

template <typename T>
__global__ __launch_bounds__(256) void hupr_k_bn_bwd_apply(const T* __restrict__ dy, const T* __restrict__ y,
                                                           const T* __restrict__ x,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ sums, float inv_m,
                                                           T* __restrict__ dx, long n4, int C, int train) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int c = (int)((i * 4) % C);
        const float4 gv = ld_act4(dy + i * 4);
        const float4 xv = ld_act4(x + i * 4);
        float g[4] = {gv.x, gv.y, gv.z, gv.w};
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
        if (y) {
            const float4 yv = ld_act4(y + i * 4);
            const float ys[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) g[k] = ys[k] > 0.f ? g[k] : 0.f;
        }
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float is = invstd[c + k], w = gamma[c + k] * is;
            if (train) {
                const float xh = (xs[k] - mean[c + k]) * is;
                o[k] = w * (g[k] - sums[c + k] * inv_m - xh * sums[C + c + k] * inv_m);
            } else {
                o[k] = w * g[k];
            }
        }
        st_act4(dx + i * 4, make_float4(o[0], o[1], o[2], o[3]));
    }
}

// ---- PReLU with one shared slope (nn.PReLU() default) --------------------------------------
__global__ __launch_bounds__(256) void hupr_k_prelu_fwd(const float* __restrict__ x, const float* __restrict__ alpha,
                                                        float* __restrict__ y, long n4) {
    const float a = alpha[0];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 v = reinterpret_cast<const float4*>(x)[i];
        v.x = v.x > 0.f ? v.x : a * v.x; v.y = v.y > 0.f ? v.y : a * v.y;
        v.z = v.z > 0.f ? v.z : a * v.z; v.w = v.w > 0.f ? v.w : a * v.w;
        reinterpret_cast<float4*>(y)[i] = v;
    }
}

// dx = dy * (x > 0 ? 1 : alpha);  partial[blk] = sum dy * x * [x <= 0]
__global__ __launch_bounds__(256) void hupr_k_prelu_bwd(const float* __restrict__ dy, const float* __restrict__ x,
                                                        const float* __restrict__ alpha, float* __restrict__ dx,
                                                        long n4, double* __restrict__ partial) {
    __shared__ double red[4];
    const float a = alpha[0];
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 g = reinterpret_cast<const float4*>(dy)[i];
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        float4 o;
        o.x = v.x > 0.f ? g.x : a * g.x; o.y = v.y > 0.f ? g.y : a * g.y;
        o.z = v.z > 0.f ? g.z : a * g.z; o.w = v.w > 0.f ? g.w : a * g.w;
        acc += (v.x > 0.f ? 0.f : g.x * v.x) + (v.y > 0.f ? 0.f : g.y * v.y) +
               (v.z > 0.f ? 0.f : g.z * v.z) + (v.w > 0.f ? 0.f : g.w * v.w);
        reinterpret_cast<float4*>(dx)[i] = o;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = (double)acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void hupr_k_sum_partials(const double* __restrict__ partial, int n, float* __restrict__ out) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)(red[0] + red[1] + red[2] + red[3]);
}

__global__ void hupr_k_colsum_final(const double* __restrict__ partial, int nblk, int C, float* __restrict__ out) {
    int c;
    double s, unused;
    if (!reduce_partials(partial, nblk, C, c, s, unused)) return;
    out[c] = (float)s;
}

static inline int ew_grid(long n4) { return (int)min((long)4096, (n4 + 255) / 256); }

}  // namespace hupr

using namespace hupr;

extern "C" size_t hupr_bn_ws_bytes(int C) { return (size_t)kStatBlocks * 2 * C * sizeof(double) + 2 * C * sizeof(float); }

static int bn_check(const char* who, long M, int C) {
    HUPR_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && C <= 1024, "%s: unsupported shape M=%ld C=%d", who, M, C);
    return HUPR_OK;
}

// (a4) BatchNorm3d, training mode: batch statistics + running-stat update (momentum), and the
// folded per-channel scale/shift used by hupr_scale_shift_act_*.  models/layers.py:46,49,53
template <typename T>
static int bn_train_stats(const char* who, const T* x, long M, int C, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float momentum, float eps, float* save_mean,
                          float* save_invstd, float* scale, float* shift, void* ws, size_t ws_bytes, hupr_stream_t stream) {
    HUPR_REQUIRE(x && gamma && beta && save_mean && save_invstd && scale && shift && ws, "%s: null pointer", who);
    int rc = bn_check(who, M, C);
    if (rc) return rc;
    if (ws_bytes < hupr_bn_ws_bytes(C)) return fail(HUPR_ERR_WORKSPACE, "%s: workspace too small", who);
    hipStream_t s = as_stream(stream);
    const int nblk = (int)min((long)kStatBlocks, (M + 63) / 64);
    double* partial = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL((hupr_k_colstats<0, T>), dim3(nblk), dim3(256), 2 * C * sizeof(double), s, x, (const T*)nullptr,
                       (const T*)nullptr, nullptr, nullptr, M, C, partial);
    HUPR_LAUNCH_OK("hupr_k_colstats<0>");
    hipLaunchKernelGGL(hupr_k_bn_finalize_fwd, dim3((C + 63) / 64), dim3(256), 0, s, partial, nblk, M, C, gamma,
                       beta, running_mean, running_var, momentum, eps, save_mean, save_invstd, scale, shift);
    HUPR_LAUNCH_OK("hupr_k_bn_finalize_fwd");
    return HUPR_OK;
}

extern "C" int hupr_bn_train_stats_f32(const float* x, long M, int C, const float* gamma, const float* beta,
                                       float* running_mean, float* running_var, float momentum, float eps,
                                       float* save_mean, float* save_invstd, float* scale, float* shift,
                                       void* ws, size_t ws_bytes, hupr_stream_t stream) {
    return bn_train_stats("hupr_bn_train_stats_f32", x, M, C, gamma, beta, running_mean, running_var, momentum, eps,
                          save_mean, save_invstd, scale, shift, ws, ws_bytes, stream);
}
extern "C" int hupr_bn_train_stats_bf16act(const void* x, long M, int C, const float* gamma, const float* beta,
                                           float* running_mean, float* running_var, float momentum, float eps,
                                           float* save_mean, float* save_invstd, float* scale, float* shift,
                                           void* ws, size_t ws_bytes, hupr_stream_t stream) {
    return bn_train_stats("hupr_bn_train_stats_bf16act", static_cast<const __bf16*>(x), M, C, gamma, beta, running_mean,
                          running_var, momentum, eps, save_mean, save_invstd, scale, shift, ws, ws_bytes, stream);
}

extern "C" int hupr_bn_eval_params_f32(const float* gamma, const float* beta, const float* running_mean,
                                       const float* running_var, float eps, int C, float* scale, float* shift,
                                       hupr_stream_t stream) {
    HUPR_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C > 0, "hupr_bn_eval_params_f32: bad argument");
    hipLaunchKernelGGL(hupr_k_bn_eval_params, dim3((C + 127) / 128), dim3(128), 0, as_stream(stream), gamma, beta,
                       running_mean, running_var, eps, C, scale, shift);
    HUPR_LAUNCH_OK("hupr_k_bn_eval_params");
    return HUPR_OK;
}

// y = act(x1*scale1 + shift1 [+ x2*scale2 + shift2]); act 0 = identity, 1 = ReLU
template <typename T>
static int scale_shift_act(const char* who, const T* x1, const float* scale1, const float* shift1, const T* x2,
                           const float* scale2, const float* shift2, T* y, long M, int C, int act, hupr_stream_t stream) {
    HUPR_REQUIRE(x1 && scale1 && shift1 && y, "%s: null pointer", who);
    HUPR_REQUIRE(!x2 || (scale2 && shift2), "%s: second branch needs scale/shift", who);
    int rc = bn_check(who, M, C);
    if (rc) return rc;
    const long n4 = M * C / 4;
    hipLaunchKernelGGL(hupr_k_scale_shift_act<T>, dim3(ew_grid(n4)), dim3(256), 0, as_stream(stream), x1, scale1, shift1,
                       x2, scale2, shift2, y, n4, C, act);
    HUPR_LAUNCH_OK("hupr_k_scale_shift_act");
    return HUPR_OK;
}
extern "C" int hupr_scale_shift_act_f32(const float* x1, const float* scale1, const float* shift1, const float* x2,
                                        const float* scale2, const float* shift2, float* y, long M, int C, int act,
                                        hupr_stream_t stream) {
    return scale_shift_act("hupr_scale_shift_act_f32", x1, scale1, shift1, x2, scale2, shift2, y, M, C, act, stream);
}
extern "C" int hupr_scale_shift_act_bf16act(const void* x1, const float* scale1, const float* shift1, const void* x2,
                                            const float* scale2, const float* shift2, void* y, long M, int C, int act,
                                            hupr_stream_t stream) {
    return scale_shift_act("hupr_scale_shift_act_bf16act", static_cast<const __bf16*>(x1), scale1, shift1,
                           static_cast<const __bf16*>(x2), scale2, shift2, static_cast<__bf16*>(y), M, C, act, stream);
}

// BatchNorm backward through an optional ReLU mask (y > 0).  train=1: batch-stat formula.
template <typename T>
static int bn_bwd(const char* who, const T* dy, const T* y_mask, const T* x, const float* save_mean,
                  const float* save_invstd, const float* gamma, T* dx, float* dgamma, float* dbeta, long M, int C,
                  int train, void* ws, size_t ws_bytes, hupr_stream_t stream) {
    HUPR_REQUIRE(dy && x && save_mean && save_invstd && gamma && dx && dgamma && dbeta && ws, "%s: null pointer", who);
    int rc = bn_check(who, M, C);
    if (rc) return rc;
    if (ws_bytes < hupr_bn_ws_bytes(C)) return fail(HUPR_ERR_WORKSPACE, "%s: workspace too small", who);
    hipStream_t s = as_stream(stream);
    const int nblk = (int)min((long)kStatBlocks, (M + 63) / 64);
    double* partial = reinterpret_cast<double*>(ws);
    float* sums = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + (size_t)kStatBlocks * 2 * C * sizeof(double));
    hipLaunchKernelGGL((hupr_k_colstats<1, T>), dim3(nblk), dim3(256), 2 * C * sizeof(double), s, x, dy, y_mask, save_mean,
                       save_invstd, M, C, partial);
    HUPR_LAUNCH_OK("hupr_k_colstats<1>");
    hipLaunchKernelGGL(hupr_k_bn_finalize_bwd, dim3((C + 63) / 64), dim3(256), 0, s, partial, nblk, C, dgamma, dbeta, sums);
    HUPR_LAUNCH_OK("hupr_k_bn_finalize_bwd");
    const long n4 = M * C / 4;
    hipLaunchKernelGGL(hupr_k_bn_bwd_apply<T>, dim3(ew_grid(n4)), dim3(256), 0, s, dy, y_mask, x, save_mean, save_invstd,
                       gamma, sums, 1.0f / (float)M, dx, n4, C, train);
    HUPR_LAUNCH_OK("hupr_k_bn_bwd_apply");
    return HUPR_OK;
}
extern "C" int hupr_bn_bwd_f32(const float* dy, const float* y_mask, const float* x, const float* save_mean,
                               const float* save_invstd, const float* gamma, float* dx, float* dgamma, float* dbeta,
                               long M, int C, int train, void* ws, size_t ws_bytes, hupr_stream_t stream) {
    return bn_bwd("hupr_bn_bwd_f32", dy, y_mask, x, save_mean, save_invstd, gamma, dx, dgamma, dbeta, M, C, train, ws,
                  ws_bytes, stream);
}
extern "C" int hupr_bn_bwd_bf16act(const void* dy, const void* y_mask, const void* x, const float* save_mean,
                                   const float* save_invstd, const float* gamma, void* dx, float* dgamma, float* dbeta,
                                   long M, int C, int train, void* ws, size_t ws_bytes, hupr_stream_t stream) {
    return bn_bwd("hupr_bn_bwd_bf16act", static_cast<const __bf16*>(dy), static_cast<const __bf16*>(y_mask),
                  static_cast<const __bf16*>(x), save_mean, save_invstd, gamma, static_cast<__bf16*>(dx), dgamma, dbeta,
                  M, C, train, ws, ws_bytes, stream);
}

extern "C" int hupr_prelu_fwd_f32(const float* x, const float* alpha, float* y, long n, hupr_stream_t stream) {
    HUPR_REQUIRE(x && alpha && y && n > 0 && n % 4 == 0, "hupr_prelu_fwd_f32: bad argument");
    hipLaunchKernelGGL(hupr_k_prelu_fwd, dim3(ew_grid(n / 4)), dim3(256), 0, as_stream(stream), x, alpha, y, n / 4);
    HUPR_LAUNCH_OK("hupr_k_prelu_fwd");
    return HUPR_OK;
}

extern "C" size_t hupr_prelu_ws_bytes(void) { return 4096 * sizeof(double); }

extern "C" int hupr_prelu_bwd_f32(const float* dy, const float* x, const float* alpha, float* dx, float* dalpha,
                                  long n, void* ws, size_t ws_bytes, hupr_stream_t stream) {
    HUPR_REQUIRE(dy && x && alpha && dx && dalpha && ws && n > 0 && n % 4 == 0, "hupr_prelu_bwd_f32: bad argument");
    if (ws_bytes < hupr_prelu_ws_bytes()) return fail(HUPR_ERR_WORKSPACE, "hupr_prelu_bwd_f32: workspace too small");
    hipStream_t s = as_stream(stream);
    const int grid = ew_grid(n / 4);
    double* partial = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL(hupr_k_prelu_bwd, dim3(grid), dim3(256), 0, s, dy, x, alpha, dx, n / 4, partial);
    HUPR_LAUNCH_OK("hupr_k_prelu_bwd");
    hipLaunchKernelGGL(hupr_k_sum_partials, dim3(1), dim3(256), 0, s, partial, grid, dalpha);
    HUPR_LAUNCH_OK("hupr_k_sum_partials");
    return HUPR_OK;
}

// out[c] = sum over rows of x[row][c]   (bias gradient of a convolution)
template <typename T>
static int colsum(const char* who, const T* x, long M, int C, float* out, void* ws, size_t ws_bytes, hupr_stream_t stream) {
    HUPR_REQUIRE(x && out && ws, "%s: null pointer", who);
    int rc = bn_check(who, M, C);
    if (rc) return rc;
    if (ws_bytes < hupr_bn_ws_bytes(C)) return fail(HUPR_ERR_WORKSPACE, "%s: workspace too small", who);
    hipStream_t s = as_stream(stream);
    const int nblk = (int)min((long)kStatBlocks, (M + 63) / 64);
    double* partial = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL((hupr_k_colstats<0, T>), dim3(nblk), dim3(256), 2 * C * sizeof(double), s, x, (const T*)nullptr,
                       (const T*)nullptr, nullptr, nullptr, M, C, partial);
    HUPR_LAUNCH_OK("hupr_k_colstats<0>");
    hipLaunchKernelGGL(hupr_k_colsum_final, dim3((C + 63) / 64), dim3(256), 0, s, partial, nblk, C, out);
    HUPR_LAUNCH_OK("hupr_k_colsum_final");
    return HUPR_OK;
}
extern "C" int hupr_colsum_f32(const float* x, long M, int C, float* out, void* ws, size_t ws_bytes, hupr_stream_t stream) {
    return colsum("hupr_colsum_f32", x, M, C, out, ws, ws_bytes, stream);
}
extern "C" int hupr_colsum_bf16act(const void* x, long M, int C, float* out, void* ws, size_t ws_bytes, hupr_stream_t stream) {
    return colsum("hupr_colsum_bf16act", static_cast<const __bf16*>(x), M, C, out, ws, ws_bytes, stream);
}

// fp32 <-> bf16 activation casts at the boundary of the bf16-activation region (n % 4 == 0)
namespace hupr {
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void hupr_k_cast(const TI* __restrict__ x, TO* __restrict__ y, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) st_act4(y + i * 4, ld_act4(x + i * 4));
}
}  // namespace hupr
extern "C" int hupr_cast_f32_to_bf16(const float* x, void* y, long n, hupr_stream_t stream) {
    HUPR_REQUIRE(x && y && n > 0 && n % 4 == 0, "hupr_cast_f32_to_bf16: bad argument");
    hipLaunchKernelGGL((hupr_k_cast<float, __bf16>), dim3(ew_grid(n / 4)), dim3(256), 0, as_stream(stream), x,
                       static_cast<__bf16*>(y), n / 4);
    HUPR_LAUNCH_OK("hupr_k_cast");
    return HUPR_OK;
}
extern "C" int hupr_cast_bf16_to_f32(const void* x, float* y, long n, hupr_stream_t stream) {
    HUPR_REQUIRE(x && y && n > 0 && n % 4 == 0, "hupr_cast_bf16_to_f32: bad argument");
    hipLaunchKernelGGL((hupr_k_cast<__bf16, float>), dim3(ew_grid(n / 4)), dim3(256), 0, as_stream(stream),
                       static_cast<const __bf16*>(x), y, n / 4);
    HUPR_LAUNCH_OK("hupr_k_cast");
    return HUPR_OK;
}
