// BatchNorm (train/eval, fwd/bwd), scale-shift-activation, PReLU — all on channels-last
// fp32 activations x[row][C] (row = (b,d,h,w) voxel).  HBM-bound elementwise/reduction kernels:
// float4 accesses, per-thread partial sums over short row runs, cross-thread combination in
// double (BatchNorm variance is formed from E[x^2]-E[x]^2, so the sums themselves must be tight).
//
// Mirrors the semantics of nn.BatchNorm3d / nn.ReLU / nn.PReLU as used by the reference's
// BasicBlock3D (models/layers.py:40-70) and BasicBlock2D (models/layers.py:8-38).
#include "hupr_common.h"

namespace hupr {

constexpr int kStatBlocks = 512;

// ------------------------------------------------------------------------------------------
// column statistics: for every channel c: S1 = sum_r f(r,c), S2 = sum_r g(r,c)
//   MODE 0 (forward) : f = x,             g = x*x
//   MODE 1 (backward): f = dy',           g = dy' * xhat      dy' = dy * [y > 0] if y given, or — fs / ft given — dy *
//                      [fs x + ft > 0]: the ReLU mask recomputed from x with the forward pass's scale / shift (the
//                      very expression hupr_k_scale_shift_act evaluated) instead of read back as a third tensor
// partial[blk][2][C] doubles.  A thread owns V channels and walks rows with four 16-byte loads per
// operand in flight (the kernel is latency-, not bandwidth-limited otherwise).
// ------------------------------------------------------------------------------------------
template <int MODE, typename T>
__global__ __launch_bounds__(256) void hupr_k_colstats(const T* __restrict__ x, const T* __restrict__ dy,
                                                       const T* __restrict__ y,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, long M, int C,
                                                       double* __restrict__ partial, const float* __restrict__ fs,
                                                       const float* __restrict__ ft) {
    constexpr int V = ActVec<T>::V, U = 4;
    // Every row subgroup leaves its fp32 column sums in its own LDS row [rsub][2][C]; they are then added in fp64 in subgroup
    // order — no atomics: the same input gives the same statistics bit for bit on every run (LDS double atomics made the order,
    // hence the last bit of a mean now and then, vary, which 120 chaotic training steps amplify: tests/test_trained_gpu.py).
    extern __shared__ float shf[];   // [rows_per_pass][2][C]
    const int tid = threadIdx.x;
    const int cvn = C / V;                       // channel groups per row
    const int rows_per_pass = 256 / cvn;         // C / V <= 256
    const int cv = tid % cvn, rsub = tid / cvn;
    const long rows_per_block = (M + gridDim.x - 1) / gridDim.x;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(M, r0 + rows_per_block);
    float s1[V], s2[V], mu[V], is[V], ma[V], mb[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { s1[k] = 0.f; s2[k] = 0.f; mu[k] = 0.f; is[k] = 1.f; ma[k] = 0.f; mb[k] = 1.f; }
    if (MODE == 1) {
#pragma unroll
        for (int k = 0; k < V; ++k) { mu[k] = mean[cv * V + k]; is[k] = invstd[cv * V + k]; }
        if (fs) {
#pragma unroll
            for (int k = 0; k < V; ++k) { ma[k] = fs[cv * V + k]; mb[k] = ft[cv * V + k]; }
        }
    }
    if (rsub < rows_per_pass) {
        for (long r = r0 + rsub; r < r1; r += (long)U * rows_per_pass) {
            float xs[U][V], gs[U][V], ys[U][V];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long ru = r + (long)u * rows_per_pass;
                if (ru < r1) {
                    ActVec<T>::load(x + ru * C + cv * V, xs[u]);
                    if (MODE == 1) {
                        ActVec<T>::load(dy + ru * C + cv * V, gs[u]);
                        if (y) ActVec<T>::load(y + ru * C + cv * V, ys[u]);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < V; ++k) { xs[u][k] = 0.f; gs[u][k] = 0.f; ys[u][k] = 1.f; }
                    if (MODE == 1) {
#pragma unroll
                        for (int k = 0; k < V; ++k) xs[u][k] = mu[k];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    if (MODE == 0) {
                        s1[k] += xs[u][k];
                        s2[k] = fmaf(xs[u][k], xs[u][k], s2[k]);
                    } else {
                        float g = (y && !(ys[u][k] > 0.f)) ? 0.f : gs[u][k];
                        if (fs && !(fmaf(xs[u][k], ma[k], mb[k]) > 0.f)) g = 0.f;
                        s1[k] += g;
                        s2[k] = fmaf(g, (xs[u][k] - mu[k]) * is[k], s2[k]);
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            shf[(rsub * 2 + 0) * C + cv * V + k] = s1[k];
            shf[(rsub * 2 + 1) * C + cv * V + k] = s2[k];
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * C; i += 256) {
        double a = 0.0;
        for (int r = 0; r < rows_per_pass; ++r) a += (double)shf[r * 2 * C + i];
        partial[(long)blockIdx.x * 2 * C + i] = a;
    }
}


// sum partial[b][2][C] over b for channel c; 256 threads = 16 row-groups x 16 channels per block
__device__ __forceinline__ bool reduce_partials(const double* __restrict__ partial, int nblk, int C, int& c,
                                                double& s1, double& s2) {
    __shared__ double sh1[16][17], sh2[16][17];
    const int g = threadIdx.x >> 4, cl = threadIdx.x & 15;
    c = blockIdx.x * 16 + cl;
    double a = 0.0, b2 = 0.0;
    if (c < C) {
#pragma unroll 4
        for (int b = g; b < nblk; b += 16) {
            a += partial[(long)b * 2 * C + c];
            b2 += partial[(long)b * 2 * C + C + c];
        }
    }
    sh1[g][cl] = a;
    sh2[g][cl] = b2;
    __syncthreads();
    s1 = 0.0;
    s2 = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { s1 += sh1[k][cl]; s2 += sh2[k][cl]; }
    return g == 0 && c < C;
}
constexpr int kFinalizeCh = 16;     // channels per finalize workgroup

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

// the two forward finalizes of a BasicBlock3D tail (both branches' convolutions left their column sums) in one launch: blockIdx.y = branch
struct BnFinalizeFwd {
    const double* partial;
    int nblk;
    const float *gamma, *beta;
    float *running_mean, *running_var, *save_mean, *save_invstd, *scale, *shift;
    float momentum, eps;
};
__global__ void hupr_k_bn_finalize_fwd2(BnFinalizeFwd a0, BnFinalizeFwd a1, long M, int C) {
    const BnFinalizeFwd& a = blockIdx.y ? a1 : a0;
    int c;
    double s1, s2;
    if (!reduce_partials(a.partial, a.nblk, C, c, s1, s2)) return;
    const double mean = s1 / (double)M;
    double var = s2 / (double)M - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)a.eps));
    a.save_mean[c] = (float)mean;
    a.save_invstd[c] = invstd;
    const float sc = a.gamma[c] * invstd;
    a.scale[c] = sc;
    a.shift[c] = a.beta[c] - (float)mean * sc;
    if (a.running_mean) {
        const double unbiased = (M > 1) ? var * (double)M / (double)(M - 1) : var;
        a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * (float)mean;
        a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)unbiased;
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

// per-thread channel coefficients: when the grid stride is a multiple of C the V channels of a thread never
// change, so the coefficient vectors are loaded once instead of once per element group
template <int V>
__device__ __forceinline__ void load_coef(const float* __restrict__ a, int c, float* v) {
#pragma unroll
    for (int k = 0; k < V; k += 4) {
        const float4 t = *reinterpret_cast<const float4*>(a + c + k);
        v[k] = t.x; v[k + 1] = t.y; v[k + 2] = t.z; v[k + 3] = t.w;
    }
}

// y = act(x1*s1 + t1 (+ x2*s2 + t2)) ; act: 0 none, 1 relu
template <typename T>
__global__ __launch_bounds__(256) void hupr_k_scale_shift_act(const T* __restrict__ x1,
                                                              const float* __restrict__ s1,
                                                              const float* __restrict__ t1,
                                                              const T* __restrict__ x2,
                                                              const float* __restrict__ s2,
                                                              const float* __restrict__ t2,
                                                              T* __restrict__ y, long nv, int C, int act) {
    constexpr int V = ActVec<T>::V;
    const long stride = (long)gridDim.x * 256;
    const bool fixed = (stride * V) % C == 0;
    float a[V], b[V], a2[V], b2[V];
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    int c = (int)((i * V) % C);
    if (fixed) {
        load_coef<V>(s1, c, a); load_coef<V>(t1, c, b);
        if (x2) { load_coef<V>(s2, c, a2); load_coef<V>(t2, c, b2); }
    }
    for (; i < nv; i += stride) {
        if (!fixed) {
            c = (int)((i * V) % C);
            load_coef<V>(s1, c, a); load_coef<V>(t1, c, b);
            if (x2) { load_coef<V>(s2, c, a2); load_coef<V>(t2, c, b2); }
        }
        float v[V], u[V];
        ActVec<T>::load(x1 + i * V, v);
        if (x2) ActVec<T>::load(x2 + i * V, u);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            v[k] = fmaf(v[k], a[k], b[k]);
            if (x2) v[k] += fmaf(u[k], a2[k], b2[k]);
            if (act == 1) v[k] = fmaxf(v[k], 0.f);
        }
        ActVec<T>::store(y + i * V, v);
    }
}

// Eval-mode BatchNorm (+ second branch) (+ ReLU) in ONE launch: the per-channel coefficients — the very expressions of
// hupr_k_bn_eval_params — are evaluated by each thread for its own channels instead of by a launch of their own in front
// (single-sample inference is launch-bound: 30 BatchNorms per forward pass).
template <int V>
__device__ __forceinline__ void eval_coef(const float* __restrict__ g, const float* __restrict__ b, const float* __restrict__ rm,
                                          const float* __restrict__ rv, float eps, int c, float* sc, float* sh) {
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const float invstd = 1.0f / sqrtf(rv[c + k] + eps);
        const float s = g[c + k] * invstd;
        sc[k] = s;
        sh[k] = b[c + k] - rm[c + k] * s;
    }
}
struct BnEvalSide { const float *gamma, *beta, *rm, *rv; float eps; };
template <typename T>
__global__ __launch_bounds__(256) void hupr_k_bn_eval_act(const T* __restrict__ x1, BnEvalSide p1, const T* __restrict__ x2,
                                                          BnEvalSide p2, T* __restrict__ y, long nv, int C, int act) {
    constexpr int V = ActVec<T>::V;
    const long stride = (long)gridDim.x * 256;
    const bool fixed = (stride * V) % C == 0;
    float a[V], b[V], a2[V], b2[V];
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    int c = (int)((i * V) % C);
    if (fixed) {
        eval_coef<V>(p1.gamma, p1.beta, p1.rm, p1.rv, p1.eps, c, a, b);
        if (x2) eval_coef<V>(p2.gamma, p2.beta, p2.rm, p2.rv, p2.eps, c, a2, b2);
    }
    for (; i < nv; i += stride) {
        if (!fixed) {
            c = (int)((i * V) % C);
            eval_coef<V>(p1.gamma, p1.beta, p1.rm, p1.rv, p1.eps, c, a, b);
            if (x2) eval_coef<V>(p2.gamma, p2.beta, p2.rm, p2.rv, p2.eps, c, a2, b2);
        }
        float v[V], u[V];
        ActVec<T>::load(x1 + i * V, v);
        if (x2) ActVec<T>::load(x2 + i * V, u);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            v[k] = fmaf(v[k], a[k], b[k]);
            if (x2) v[k] += fmaf(u[k], a2[k], b2[k]);
            if (act == 1) v[k] = fmaxf(v[k], 0.f);
        }
        ActVec<T>::store(y + i * V, v);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Inference tails on K-SLICED convolutions (config C2: single-sample forward; conv_halo_bf16.hip, HaloArgs::part).  A sliced
// convolution leaves n fp32 partial tensors [n][M][C]; instead of a reduce launch per convolution, the elementwise kernel that
// consumes the convolution sums the slices itself — in slice order, rounded to bf16 exactly where the stored tensor would have
// been rounded, so the results are bit-identical to convolution -> reduce -> tail.
//   MODE 0  BasicBlock3D:  y = relu?( bn1_eval(c1) [+ bn2_eval(c2)] )                 (models/layers.py:55-70, eval mode)
//   MODE 1  BasicBlock2D:  y = prelu( c1 [+ c2] )   with c1 + c2 rounded once, as the convolution's residual epilogue does
// A side is either a bf16 tensor (n == 0) or n fp32 slices.  4 channels per thread.
// ------------------------------------------------------------------------------------------------------------------
struct TailSide { const void* x; int n; BnEvalSide bn; };
__device__ __forceinline__ void tail_load(const TailSide& s, long i4, long MC, float* v, bool round_it) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    if (s.n == 0) {
        const bf16x4 t = *reinterpret_cast<const bf16x4*>(static_cast<const __bf16*>(s.x) + i4 * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (float)t[k];
        return;
    }
    const float* p = static_cast<const float*>(s.x) + i4 * 4;
    float4 a = *reinterpret_cast<const float4*>(p);
    for (int j = 1; j < s.n; ++j) {
        const float4 t = *reinterpret_cast<const float4*>(p + (long)j * MC);
        a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
    }
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    if (round_it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (float)(__bf16)v[k];
    }
}
template <int MODE>
__global__ __launch_bounds__(256) void hupr_k_infer_tail(TailSide s1, TailSide s2, const float* __restrict__ alpha, int relu,
                                                         __bf16* __restrict__ y, long M, int C) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    const long MC = M * C, n4 = MC >> 2;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)((i * 4) % C);
    float v[4], u[4];
    if (MODE == 0) {
        float a[4], b[4];
        tail_load(s1, i, MC, v, true);
        eval_coef<4>(s1.bn.gamma, s1.bn.beta, s1.bn.rm, s1.bn.rv, s1.bn.eps, c, a, b);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = fmaf(v[k], a[k], b[k]);
        if (s2.x) {
            tail_load(s2, i, MC, u, true);
            eval_coef<4>(s2.bn.gamma, s2.bn.beta, s2.bn.rm, s2.bn.rv, s2.bn.eps, c, a, b);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += fmaf(u[k], a[k], b[k]);
        }
        if (relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
        }
    } else {
        tail_load(s1, i, MC, v, false);
        if (s2.x) {
            tail_load(s2, i, MC, u, true);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += u[k];
        }
        const float al = alpha[0];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = (float)(__bf16)v[k];                       // the convolution's stored output
            v[k] = v[k] > 0.f ? v[k] : al * v[k];
        }
    }
    const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    *reinterpret_cast<bf16x4*>(y + i * 4) = o;
}

// backward finalize: dgamma = S2, dbeta = S1 and the per-channel coefficients of the apply pass
//   train: dx = w*(g' - S1/M - xhat*S2/M) = cA*g' + cB*(x - mean) + cD,  cA = w, cB = -w*invstd*S2/M, cD = -w*S1/M
//   eval : dx = w*g'                                                      cB = cD = 0          (w = gamma*invstd)
__global__ void hupr_k_bn_finalize_bwd(const double* __restrict__ partial, int nblk, int C,
                                       const float* __restrict__ gamma, const float* __restrict__ invstd, float inv_m,
                                       int train, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ coef /* [3][C] floats */) {
    int c;
    double s1, s2;
    if (!reduce_partials(partial, nblk, C, c, s1, s2)) return;
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
    const float is = invstd[c], w = gamma[c] * is;
    coef[c] = w;
    coef[C + c] = train ? -w * is * ((float)s2 * inv_m) : 0.f;
    coef[2 * C + c] = train ? -w * ((float)s1 * inv_m) : 0.f;
}

template <typename T>
__global__ __launch_bounds__(256) void hupr_k_bn_bwd_apply(const T* __restrict__ dy, const T* __restrict__ y,
                                                           const T* __restrict__ x,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ coef,
                                                           T* __restrict__ dx, long nv, int C,
                                                           const float* __restrict__ fs, const float* __restrict__ ft) {
    constexpr int V = ActVec<T>::V;
    const long stride = (long)gridDim.x * 256;
    const bool fixed = (stride * V) % C == 0;
    float cA[V], cB[V], cD[V], mu[V], ma[V], mb[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { ma[k] = 0.f; mb[k] = 1.f; }
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    int c = (int)((i * V) % C);
#define HUPR_LOAD_COEFS()                                                                                           \
    load_coef<V>(coef, c, cA); load_coef<V>(coef + C, c, cB); load_coef<V>(coef + 2 * C, c, cD); load_coef<V>(mean, c, mu); \
    if (fs) { load_coef<V>(fs, c, ma); load_coef<V>(ft, c, mb); }
    if (fixed) { HUPR_LOAD_COEFS() }
    for (; i < nv; i += stride) {
        if (!fixed) {
            c = (int)((i * V) % C);
            HUPR_LOAD_COEFS()
        }
        float g[V], xs[V], ys[V], o[V];
        ActVec<T>::load(dy + i * V, g);
        ActVec<T>::load(x + i * V, xs);
        if (y) ActVec<T>::load(y + i * V, ys);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float gm = (y && !(ys[k] > 0.f)) ? 0.f : g[k];
            if (fs && !(fmaf(xs[k], ma[k], mb[k]) > 0.f)) gm = 0.f;
            o[k] = fmaf(cA[k], gm, fmaf(cB[k], xs[k] - mu[k], cD[k]));
        }
        ActVec<T>::store(dx + i * V, o);
    }
#undef HUPR_LOAD_COEFS
}

// ------------------------------------------------------------------------------------------
// Two-branch variant for the BasicBlock3D tail y = relu(bn_a(x1) + bn_b(x2)): both branches share dy and the ReLU mask,
// so one statistics pass (S1 = sum dy', S2a = sum dy' xhat1, S2b = sum dy' xhat2; partial[blk][3][C]) and one apply pass
// (dx1, dx2) replace two of each — 4 + 6 tensor reads/writes instead of 6 + 8.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void hupr_k_colstats2(const T* __restrict__ x1, const T* __restrict__ x2,
                                                        const T* __restrict__ dy, const T* __restrict__ y,
                                                        const float* __restrict__ mean1, const float* __restrict__ invstd1,
                                                        const float* __restrict__ mean2, const float* __restrict__ invstd2,
                                                        long M, int C, double* __restrict__ partial,
                                                        const float* __restrict__ fs1, const float* __restrict__ ft1,
                                                        const float* __restrict__ fs2, const float* __restrict__ ft2) {
    // y == null: the ReLU mask is recomputed as [fs1 x1 + ft1 + (fs2 x2 + ft2) > 0], the forward pass's own expression
    constexpr int V = ActVec<T>::V, U = 2;
    extern __shared__ float shf[];   // [rows_per_pass][3][C] (see hupr_k_colstats: deterministic, no atomics)
    const int tid = threadIdx.x;
    const int cvn = C / V;
    const int rows_per_pass = 256 / cvn;
    const int cv = tid % cvn, rsub = tid / cvn;
    const long rows_per_block = (M + gridDim.x - 1) / gridDim.x;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(M, r0 + rows_per_block);
    float s1[V], sa[V], sb[V], mu1[V], is1[V], mu2[V], is2[V], ma1[V], mb1[V], ma2[V], mb2[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
        s1[k] = sa[k] = sb[k] = 0.f;
        mu1[k] = mean1[cv * V + k]; is1[k] = invstd1[cv * V + k];
        mu2[k] = mean2[cv * V + k]; is2[k] = invstd2[cv * V + k];
        ma1[k] = ma2[k] = 0.f; mb1[k] = mb2[k] = 1.f;
        if (!y) { ma1[k] = fs1[cv * V + k]; mb1[k] = ft1[cv * V + k]; ma2[k] = fs2[cv * V + k]; mb2[k] = ft2[cv * V + k]; }
    }
    if (rsub < rows_per_pass) {
        for (long r = r0 + rsub; r < r1; r += (long)U * rows_per_pass) {
            float xa[U][V], xb[U][V], gs[U][V], ys[U][V];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long ru = r + (long)u * rows_per_pass;
                if (ru < r1) {
                    ActVec<T>::load(x1 + ru * C + cv * V, xa[u]);
                    ActVec<T>::load(x2 + ru * C + cv * V, xb[u]);
                    ActVec<T>::load(dy + ru * C + cv * V, gs[u]);
                    if (y) ActVec<T>::load(y + ru * C + cv * V, ys[u]);
                } else {
#pragma unroll
                    for (int k = 0; k < V; ++k) { xa[u][k] = mu1[k]; xb[u][k] = mu2[k]; gs[u][k] = 0.f; ys[u][k] = 1.f; }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    float act = fmaf(xa[u][k], ma1[k], mb1[k]);
                    act += fmaf(xb[u][k], ma2[k], mb2[k]);
                    const float g = ((y ? ys[u][k] : act) > 0.f) ? gs[u][k] : 0.f;
                    s1[k] += g;
                    sa[k] = fmaf(g, (xa[u][k] - mu1[k]) * is1[k], sa[k]);
                    sb[k] = fmaf(g, (xb[u][k] - mu2[k]) * is2[k], sb[k]);
                }
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            shf[(rsub * 3 + 0) * C + cv * V + k] = s1[k];
            shf[(rsub * 3 + 1) * C + cv * V + k] = sa[k];
            shf[(rsub * 3 + 2) * C + cv * V + k] = sb[k];
        }
    }
    __syncthreads();
    for (int i = tid; i < 3 * C; i += 256) {
        double a = 0.0;
        for (int r = 0; r < rows_per_pass; ++r) a += (double)shf[r * 3 * C + i];
        partial[(long)blockIdx.x * 3 * C + i] = a;
    }
}

// coef[6][C]: {cA, cB, cD} of branch 1, then of branch 2 (see hupr_k_bn_finalize_bwd)
__global__ void hupr_k_bn_finalize_bwd2(const double* __restrict__ partial, int nblk, int C, const float* __restrict__ gamma1,
                                        const float* __restrict__ invstd1, const float* __restrict__ gamma2,
                                        const float* __restrict__ invstd2, float inv_m, int train, float* __restrict__ dgamma1,
                                        float* __restrict__ dbeta1, float* __restrict__ dgamma2, float* __restrict__ dbeta2,
                                        float* __restrict__ coef) {
    __shared__ double sh[3][16][17];
    const int g = threadIdx.x >> 4, cl = threadIdx.x & 15;
    const int c = blockIdx.x * 16 + cl;
    double a = 0.0, b = 0.0, d = 0.0;
    if (c < C) {
#pragma unroll 4
        for (int k = g; k < nblk; k += 16) {
            a += partial[(long)k * 3 * C + c];
            b += partial[(long)k * 3 * C + C + c];
            d += partial[(long)k * 3 * C + 2 * C + c];
        }
    }
    sh[0][g][cl] = a; sh[1][g][cl] = b; sh[2][g][cl] = d;
    __syncthreads();
    if (g != 0 || c >= C) return;
    double s1 = 0.0, sa = 0.0, sb = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { s1 += sh[0][k][cl]; sa += sh[1][k][cl]; sb += sh[2][k][cl]; }
    dbeta1[c] = (float)s1;
    dbeta2[c] = (float)s1;
    dgamma1[c] = (float)sa;
    dgamma2[c] = (float)sb;
    const float i1 = invstd1[c], w1 = gamma1[c] * i1, i2 = invstd2[c], w2 = gamma2[c] * i2;
    coef[c] = w1;
    coef[C + c] = train ? -w1 * i1 * ((float)sa * inv_m) : 0.f;
    coef[2 * C + c] = train ? -w1 * ((float)s1 * inv_m) : 0.f;
    coef[3 * C + c] = w2;
    coef[4 * C + c] = train ? -w2 * i2 * ((float)sb * inv_m) : 0.f;
    coef[5 * C + c] = train ? -w2 * ((float)s1 * inv_m) : 0.f;
}

template <typename T>
__global__ __launch_bounds__(256) void hupr_k_bn_bwd_apply2(const T* __restrict__ dy, const T* __restrict__ y,
                                                            const T* __restrict__ x1, const T* __restrict__ x2,
                                                            const float* __restrict__ mean1, const float* __restrict__ mean2,
                                                            const float* __restrict__ coef, T* __restrict__ dx1,
                                                            T* __restrict__ dx2, long nv, int C,
                                                            const float* __restrict__ fs1, const float* __restrict__ ft1,
                                                            const float* __restrict__ fs2, const float* __restrict__ ft2) {
    constexpr int V = ActVec<T>::V;
    const long stride = (long)gridDim.x * 256;
    const bool fixed = (stride * V) % C == 0;
    float a1[V], b1[V], d1[V], m1[V], a2[V], b2[V], d2[V], m2[V], ma1[V], mb1[V], ma2[V], mb2[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { ma1[k] = ma2[k] = 0.f; mb1[k] = mb2[k] = 1.f; }
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    int c = (int)((i * V) % C);
#define HUPR_LOAD_COEFS()                                                                                           \
    load_coef<V>(coef, c, a1); load_coef<V>(coef + C, c, b1); load_coef<V>(coef + 2 * C, c, d1); load_coef<V>(mean1, c, m1); \
    load_coef<V>(coef + 3 * C, c, a2); load_coef<V>(coef + 4 * C, c, b2); load_coef<V>(coef + 5 * C, c, d2); load_coef<V>(mean2, c, m2); \
    if (!y) { load_coef<V>(fs1, c, ma1); load_coef<V>(ft1, c, mb1); load_coef<V>(fs2, c, ma2); load_coef<V>(ft2, c, mb2); }
    if (fixed) { HUPR_LOAD_COEFS() }
    for (; i < nv; i += stride) {
        if (!fixed) {
            c = (int)((i * V) % C);
            HUPR_LOAD_COEFS()
        }
        float g[V], xa[V], xb[V], ys[V], o1[V], o2[V];
        ActVec<T>::load(dy + i * V, g);
        if (y) ActVec<T>::load(y + i * V, ys);
        ActVec<T>::load(x1 + i * V, xa);
        ActVec<T>::load(x2 + i * V, xb);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float act = fmaf(xa[k], ma1[k], mb1[k]);
            act += fmaf(xb[k], ma2[k], mb2[k]);
            const float gm = ((y ? ys[k] : act) > 0.f) ? g[k] : 0.f;
            o1[k] = fmaf(a1[k], gm, fmaf(b1[k], xa[k] - m1[k], d1[k]));
            o2[k] = fmaf(a2[k], gm, fmaf(b2[k], xb[k] - m2[k], d2[k]));
        }
        ActVec<T>::store(dx1 + i * V, o1);
        ActVec<T>::store(dx2 + i * V, o2);
    }
#undef HUPR_LOAD_COEFS
}

// ---- PReLU with one shared slope (nn.PReLU() default) --------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void hupr_k_prelu_fwd(const T* __restrict__ x, const float* __restrict__ alpha,
                                                        T* __restrict__ y, long nv) {
    constexpr int V = ActVec<T>::V;
    const float a = alpha[0];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        float v[V];
        ActVec<T>::load(x + i * V, v);
#pragma unroll
        for (int k = 0; k < V; ++k) v[k] = v[k] > 0.f ? v[k] : a * v[k];
        ActVec<T>::store(y + i * V, v);
    }
}

// dx = dy * (x > 0 ? 1 : alpha);  partial[blk] = sum dy * x * [x <= 0]
template <typename T>
__global__ __launch_bounds__(256) void hupr_k_prelu_bwd(const T* __restrict__ dy, const T* __restrict__ x,
                                                        const float* __restrict__ alpha, T* __restrict__ dx,
                                                        long nv, double* __restrict__ partial) {
    constexpr int V = ActVec<T>::V;
    __shared__ double red[4];
    const float a = alpha[0];
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        float g[V], v[V], o[V];
        ActVec<T>::load(dy + i * V, g);
        ActVec<T>::load(x + i * V, v);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            o[k] = v[k] > 0.f ? g[k] : a * g[k];
            acc += v[k] > 0.f ? 0.f : g[k] * v[k];
        }
        ActVec<T>::store(dx + i * V, o);
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

// several sums of partials in one launch (blockIdx.x = item): the PReLU slope gradients of a backward pass, deferred until their
// gradient bucket is complete (tools/distributed.py) — each item summed exactly as hupr_k_sum_partials sums it
struct SumItems {
    int n_items;
    const double* partial[16];
    int n[16];
    float* out[16];
};
__global__ void hupr_k_sum_partials_multi(SumItems t) {
    __shared__ double red[4];
    const int it = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < t.n[it]; i += 256) s += t.partial[it][i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) t.out[it][0] = (float)(red[0] + red[1] + red[2] + red[3]);
}

__global__ void hupr_k_colsum_final(const double* __restrict__ partial, int nblk, int C, float* __restrict__ out) {
    int c;
    double s, unused;
    if (!reduce_partials(partial, nblk, C, c, s, unused)) return;
    out[c] = (float)s;
}

static inline int ew_grid(long n4) { return (int)min((long)4096, (n4 + 255) / 256); }
template <typename T> static inline int act_v() { return sizeof(T) == 2 ? 8 : 4; }
// dynamic LDS of the statistics kernels: [row subgroups = 256 / (C / V)][NS sums][C] floats
template <typename T> static inline size_t stats_lds(int C, int ns) { return (size_t)(256 / (C / act_v<T>())) * ns * C * sizeof(float); }

}  // namespace hupr

using namespace hupr;

extern "C" size_t hupr_bn_ws_bytes(int C) { return (size_t)kStatBlocks * 3 * C * sizeof(double) + 8 * C * sizeof(float); }

static int bn_check(const char* who, long M, int C, int V = 4) {
    HUPR_REQUIRE(M > 0 && C > 0 && C % V == 0 && C <= 1024, "%s: unsupported shape M=%ld C=%d (C must be a multiple of %d)", who, M, C, V);
    return HUPR_OK;
}

// (a4) BatchNorm3d, training mode: batch statistics + running-stat update (momentum), and the
// folded per-channel scale/shift used by hupr_scale_shift_act_*.  models/layers.py:46,49,53
template <typename T>
static int bn_train_stats(const char* who, const T* x, long M, int C, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float momentum, float eps, float* save_mean,
                          float* save_invstd, float* scale, float* shift, void* ws, size_t ws_bytes, hupr_stream_t stream) {
    HUPR_REQUIRE(x && gamma && beta && save_mean && save_invstd && scale && shift && ws, "%s: null pointer", who);
    int rc = bn_check(who, M, C, act_v<T>());
    if (rc) return rc;
    if (ws_bytes < hupr_bn_ws_bytes(C)) return fail(HUPR_ERR_WORKSPACE, "%s: workspace too small", who);
    hipStream_t s = as_stream(stream);
    const int nblk = (int)min((long)kStatBlocks, (M + 63) / 64);
    double* partial = reinterpret_cast<double*>(ws);
    HUPR_LAUNCH((hupr_k_colstats<0, T>), dim3(nblk), dim3(256), stats_lds<T>(C, 2), s, x, (const T*)nullptr,
                       (const T*)nullptr, nullptr, nullptr, M, C, partial, (const float*)nullptr, (const float*)nullptr);
    HUPR_LAUNCH_OK("hupr_k_colstats<0>");
    HUPR_LAUNCH(hupr_k_bn_finalize_fwd, dim3((C + kFinalizeCh - 1) / kFinalizeCh), dim3(256), 0, s, partial, nblk, M, C, gamma,
                       beta, running_mean, running_var, momentum, eps, save_mean, save_invstd, scale, shift);
    HUPR_LAUNCH_OK("hupr_k_bn_finalize_fwd");
    return HUPR_OK;
}

// Finalize only: `partial` = nblk rows of [2][C] doubles (column sums and sums of squares) produced elsewhere — by the
// convolution epilogue (hupr_conv3x3_halo_bf16act_stats).  Same outputs and running-statistics update as
// hupr_bn_train_stats_*.
extern "C" int hupr_bn_train_finalize_f32(const void* partial, int nblk, long M, int C, const float* gamma, const float* beta,
                                          float* running_mean, float* running_var, float momentum, float eps,
                                          float* save_mean, float* save_invstd, float* scale, float* shift,
                                          hupr_stream_t stream) {
    HUPR_REQUIRE(partial && nblk > 0 && M > 0 && C > 0 && gamma && beta && save_mean && save_invstd && scale && shift,
                 "hupr_bn_train_finalize_f32: bad argument");
    HUPR_LAUNCH(hupr_k_bn_finalize_fwd, dim3((C + kFinalizeCh - 1) / kFinalizeCh), dim3(256), 0, as_stream(stream),
                       static_cast<const double*>(partial), nblk, M, C, gamma, beta, running_mean, running_var, momentum, eps,
                       save_mean, save_invstd, scale, shift);
    HUPR_LAUNCH_OK("hupr_k_bn_finalize_fwd");
    return HUPR_OK;
}

// ... for two BatchNorms over tensors of one shape at once (the tail of a BasicBlock3D: bn_a(conv2(...)) + bn_b(residual conv(x)),
// reference models/layers.py:66-70): one launch instead of two
extern "C" int hupr_bn_train_finalize2_f32(const void* partial1, int nblk1, const float* gamma1, const float* beta1,
                                           float* running_mean1, float* running_var1, float momentum1, float eps1,
                                           float* save_mean1, float* save_invstd1, float* scale1, float* shift1,
                                           const void* partial2, int nblk2, const float* gamma2, const float* beta2,
                                           float* running_mean2, float* running_var2, float momentum2, float eps2,
                                           float* save_mean2, float* save_invstd2, float* scale2, float* shift2, long M, int C,
                                           hupr_stream_t stream) {
    HUPR_REQUIRE(partial1 && nblk1 > 0 && gamma1 && beta1 && save_mean1 && save_invstd1 && scale1 && shift1 && partial2 && nblk2 > 0 &&
                     gamma2 && beta2 && save_mean2 && save_invstd2 && scale2 && shift2 && M > 0 && C > 0,
                 "hupr_bn_train_finalize2_f32: bad argument");
    const BnFinalizeFwd a0{static_cast<const double*>(partial1), nblk1, gamma1, beta1, running_mean1, running_var1, save_mean1,
                           save_invstd1, scale1, shift1, momentum1, eps1};
    const BnFinalizeFwd a1{static_cast<const double*>(partial2), nblk2, gamma2, beta2, running_mean2, running_var2, save_mean2,
                           save_invstd2, scale2, shift2, momentum2, eps2};
    HUPR_LAUNCH(hupr_k_bn_finalize_fwd2, dim3((C + kFinalizeCh - 1) / kFinalizeCh, 2), dim3(256), 0, as_stream(stream), a0, a1, M, C);
    HUPR_LAUNCH_OK("hupr_k_bn_finalize_fwd2");
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
    HUPR_LAUNCH(hupr_k_bn_eval_params, dim3((C + 127) / 128), dim3(128), 0, as_stream(stream), gamma, beta,
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
    int rc = bn_check(who, M, C, act_v<T>());
    if (rc) return rc;
    const long nv = M * C / act_v<T>();
    HUPR_LAUNCH(hupr_k_scale_shift_act<T>, dim3(ew_grid(nv)), dim3(256), 0, as_stream(stream), x1, scale1, shift1,
                       x2, scale2, shift2, y, nv, C, act);
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

template <typename T>
static int bn_eval_act(const char* who, const T* x1, const float* g1, const float* b1, const float* rm1, const float* rv1, float eps1,
                       const T* x2, const float* g2, const float* b2, const float* rm2, const float* rv2, float eps2, T* y, long M,
                       int C, int act, hupr_stream_t stream) {
    HUPR_REQUIRE(x1 && g1 && b1 && rm1 && rv1 && y, "%s: null pointer", who);
    HUPR_REQUIRE(!x2 || (g2 && b2 && rm2 && rv2), "%s: second branch needs its BatchNorm tensors", who);
    int rc = bn_check(who, M, C, act_v<T>());
    if (rc) return rc;
    const long nv = M * C / act_v<T>();
    HUPR_LAUNCH(hupr_k_bn_eval_act<T>, dim3(ew_grid(nv)), dim3(256), 0, as_stream(stream), x1,
                       BnEvalSide{g1, b1, rm1, rv1, eps1}, x2, BnEvalSide{g2, b2, rm2, rv2, eps2}, y, nv, C, act);
    HUPR_LAUNCH_OK("hupr_k_bn_eval_act");
    return HUPR_OK;
}
// y = act(bn1_eval(x1) [+ bn2_eval(x2)]) from the BatchNorm tensors themselves (running statistics): hupr_bn_eval_params_f32 +
// hupr_scale_shift_act_* in one launch, same arithmetic
extern "C" int hupr_bn_eval_act_f32(const float* x1, const float* gamma1, const float* beta1, const float* mean1, const float* var1,
                                    float eps1, const float* x2, const float* gamma2, const float* beta2, const float* mean2,
                                    const float* var2, float eps2, float* y, long M, int C, int act, hupr_stream_t stream) {
    return bn_eval_act("hupr_bn_eval_act_f32", x1, gamma1, beta1, mean1, var1, eps1, x2, gamma2, beta2, mean2, var2, eps2, y, M, C,
                       act, stream);
}
extern "C" int hupr_bn_eval_act_bf16act(const void* x1, const float* gamma1, const float* beta1, const float* mean1,
                                        const float* var1, float eps1, const void* x2, const float* gamma2, const float* beta2,
                                        const float* mean2, const float* var2, float eps2, void* y, long M, int C, int act,
                                        hupr_stream_t stream) {
    return bn_eval_act("hupr_bn_eval_act_bf16act", static_cast<const __bf16*>(x1), gamma1, beta1, mean1, var1, eps1,
                       static_cast<const __bf16*>(x2), gamma2, beta2, mean2, var2, eps2, static_cast<__bf16*>(y), M, C, act, stream);
}

// Inference tails that sum K-sliced convolution partials themselves (see hupr_k_infer_tail).  x*: bf16 tensor (n* == 0) or n* fp32
// slices [n][M][C]; x2 may be null.  mode 0: relu?(bn1(x1) [+ bn2(x2)]) from the BatchNorm tensors (running statistics);
// mode 1: prelu(x1 [+ x2]) with *alpha (the BatchNorm pointers are ignored).
extern "C" int hupr_infer_tail_bf16act(int mode, const void* x1, int n1, const float* g1, const float* b1, const float* m1,
                                       const float* v1, float eps1, const void* x2, int n2, const float* g2, const float* b2,
                                       const float* m2, const float* v2, float eps2, const float* alpha, int relu, void* y, long M,
                                       int C, hupr_stream_t stream) {
    const char* who = "hupr_infer_tail_bf16act";
    HUPR_REQUIRE(x1 && y && M > 0 && C > 0 && C % 4 == 0 && n1 >= 0 && n2 >= 0, "%s: bad argument", who);
    HUPR_REQUIRE(((uintptr_t)x1 & 7) == 0 && ((uintptr_t)x2 & 7) == 0 && ((uintptr_t)y & 7) == 0, "%s: misaligned pointer", who);
    HUPR_REQUIRE((n1 == 0 || ((uintptr_t)x1 & 15) == 0) && (n2 == 0 || ((uintptr_t)x2 & 15) == 0), "%s: misaligned slices", who);
    const TailSide s1{x1, n1, BnEvalSide{g1, b1, m1, v1, eps1}}, s2{x2, n2, BnEvalSide{g2, b2, m2, v2, eps2}};
    const long n4 = M * C / 4;
    const dim3 grid((unsigned)((n4 + 255) / 256));
    if (mode == 0) {
        HUPR_REQUIRE(g1 && b1 && m1 && v1 && (!x2 || (g2 && b2 && m2 && v2)), "%s: mode 0 needs the BatchNorm tensors", who);
        HUPR_LAUNCH(hupr_k_infer_tail<0>, grid, dim3(256), 0, as_stream(stream), s1, s2, alpha, relu, static_cast<__bf16*>(y), M, C);
    } else {
        HUPR_REQUIRE(mode == 1 && alpha, "%s: mode 1 needs the PReLU slope", who);
        HUPR_LAUNCH(hupr_k_infer_tail<1>, grid, dim3(256), 0, as_stream(stream), s1, s2, alpha, relu, static_cast<__bf16*>(y), M, C);
    }
    HUPR_LAUNCH_OK("hupr_k_infer_tail");
    return HUPR_OK;
}

// BatchNorm backward through an optional ReLU mask (y > 0).  train=1: batch-stat formula.
template <typename T>
static int bn_bwd(const char* who, const T* dy, const T* y_mask, const float* fs, const float* ft, const T* x,
                  const float* save_mean, const float* save_invstd, const float* gamma, T* dx, float* dgamma, float* dbeta,
                  long M, int C, int train, void* ws, size_t ws_bytes, hupr_stream_t stream) {
    HUPR_REQUIRE(dy && x && save_mean && save_invstd && gamma && dx && dgamma && dbeta && ws, "%s: null pointer", who);
    HUPR_REQUIRE((fs == nullptr) == (ft == nullptr) && !(fs && y_mask), "%s: give y_mask OR the forward scale/shift pair", who);
    int rc = bn_check(who, M, C, act_v<T>());
    if (rc) return rc;
    if (ws_bytes < hupr_bn_ws_bytes(C)) return fail(HUPR_ERR_WORKSPACE, "%s: workspace too small", who);
    hipStream_t s = as_stream(stream);
    const int nblk = (int)min((long)kStatBlocks, (M + 63) / 64);
    double* partial = reinterpret_cast<double*>(ws);
    float* coef = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + (size_t)kStatBlocks * 2 * C * sizeof(double));
    HUPR_LAUNCH((hupr_k_colstats<1, T>), dim3(nblk), dim3(256), stats_lds<T>(C, 2), s, x, dy, y_mask, save_mean,
                       save_invstd, M, C, partial, fs, ft);
    HUPR_LAUNCH_OK("hupr_k_colstats<1>");
    HUPR_LAUNCH(hupr_k_bn_finalize_bwd, dim3((C + kFinalizeCh - 1) / kFinalizeCh), dim3(256), 0, s, partial, nblk, C,
                       gamma, save_invstd, 1.0f / (float)M, train, dgamma, dbeta, coef);
    HUPR_LAUNCH_OK("hupr_k_bn_finalize_bwd");
    const long nv = M * C / act_v<T>();
    HUPR_LAUNCH(hupr_k_bn_bwd_apply<T>, dim3(ew_grid(nv)), dim3(256), 0, s, dy, y_mask, x, save_mean, coef, dx, nv, C, fs, ft);
    HUPR_LAUNCH_OK("hupr_k_bn_bwd_apply");
    return HUPR_OK;
}
extern "C" int hupr_bn_bwd_f32(const float* dy, const float* y_mask, const float* x, const float* save_mean,
                               const float* save_invstd, const float* gamma, float* dx, float* dgamma, float* dbeta,
                               long M, int C, int train, void* ws, size_t ws_bytes, hupr_stream_t stream) {
    return bn_bwd("hupr_bn_bwd_f32", dy, y_mask, nullptr, nullptr, x, save_mean, save_invstd, gamma, dx, dgamma, dbeta, M, C, train,
                  ws, ws_bytes, stream);
}
// ... with the ReLU mask recomputed from x and the forward pass's scale / shift (hupr_bn_train_stats / eval_params) instead
// of read from the activation: one tensor read less in each of the two passes
extern "C" int hupr_bn_bwd_remask_f32(const float* dy, const float* fwd_scale, const float* fwd_shift, const float* x,
                                      const float* save_mean, const float* save_invstd, const float* gamma, float* dx,
                                      float* dgamma, float* dbeta, long M, int C, int train, void* ws, size_t ws_bytes,
                                      hupr_stream_t stream) {
    HUPR_REQUIRE(fwd_scale && fwd_shift, "hupr_bn_bwd_remask_f32: null pointer");
    return bn_bwd("hupr_bn_bwd_remask_f32", dy, (const float*)nullptr, fwd_scale, fwd_shift, x, save_mean, save_invstd, gamma, dx,
                  dgamma, dbeta, M, C, train, ws, ws_bytes, stream);
}
extern "C" int hupr_bn_bwd_remask_bf16act(const void* dy, const float* fwd_scale, const float* fwd_shift, const void* x,
                                          const float* save_mean, const float* save_invstd, const float* gamma, void* dx,
                                          float* dgamma, float* dbeta, long M, int C, int train, void* ws, size_t ws_bytes,
                                          hupr_stream_t stream) {
    HUPR_REQUIRE(fwd_scale && fwd_shift, "hupr_bn_bwd_remask_bf16act: null pointer");
    return bn_bwd("hupr_bn_bwd_remask_bf16act", static_cast<const __bf16*>(dy), (const __bf16*)nullptr, fwd_scale, fwd_shift,
                  static_cast<const __bf16*>(x), save_mean, save_invstd, gamma, static_cast<__bf16*>(dx), dgamma, dbeta, M, C,
                  train, ws, ws_bytes, stream);
}
extern "C" int hupr_bn_bwd_bf16act(const void* dy, const void* y_mask, const void* x, const float* save_mean,
                                   const float* save_invstd, const float* gamma, void* dx, float* dgamma, float* dbeta,
                                   long M, int C, int train, void* ws, size_t ws_bytes, hupr_stream_t stream) {
    return bn_bwd("hupr_bn_bwd_bf16act", static_cast<const __bf16*>(dy), static_cast<const __bf16*>(y_mask), nullptr, nullptr,
                  static_cast<const __bf16*>(x), save_mean, save_invstd, gamma, static_cast<__bf16*>(dx), dgamma, dbeta,
                  M, C, train, ws, ws_bytes, stream);
}

// BatchNorm backward of y = relu(bn_a(x1) + bn_b(x2)) for both branches at once (shared dy and ReLU mask y).
template <typename T>
static int bn_bwd2(const char* who, const T* dy, const T* y_mask, const float* const* fwd /* {s1,t1,s2,t2} or null */,
                   const T* x1, const float* mean1, const float* invstd1,
                   const float* gamma1, const T* x2, const float* mean2, const float* invstd2, const float* gamma2, T* dx1,
                   T* dx2, float* dgamma1, float* dbeta1, float* dgamma2, float* dbeta2, long M, int C, int train, void* ws,
                   size_t ws_bytes, hupr_stream_t stream) {
    const float* fs1 = fwd ? fwd[0] : nullptr; const float* ft1 = fwd ? fwd[1] : nullptr;
    const float* fs2 = fwd ? fwd[2] : nullptr; const float* ft2 = fwd ? fwd[3] : nullptr;
    HUPR_REQUIRE((y_mask != nullptr) != (fs1 && ft1 && fs2 && ft2), "%s: give y_mask OR the two forward scale/shift pairs", who);
    HUPR_REQUIRE(dy && x1 && x2 && mean1 && invstd1 && gamma1 && mean2 && invstd2 && gamma2 && dx1 && dx2 && dgamma1 &&
                 dbeta1 && dgamma2 && dbeta2 && ws, "%s: null pointer", who);
    int rc = bn_check(who, M, C, act_v<T>());
    if (rc) return rc;
    if (ws_bytes < hupr_bn_ws_bytes(C)) return fail(HUPR_ERR_WORKSPACE, "%s: workspace too small", who);
    hipStream_t s = as_stream(stream);
    const int nblk = (int)min((long)kStatBlocks, (M + 63) / 64);
    double* partial = reinterpret_cast<double*>(ws);
    float* coef = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + (size_t)kStatBlocks * 3 * C * sizeof(double));
    HUPR_LAUNCH(hupr_k_colstats2<T>, dim3(nblk), dim3(256), stats_lds<T>(C, 3), s, x1, x2, dy, y_mask, mean1, invstd1,
                       mean2, invstd2, M, C, partial, fs1, ft1, fs2, ft2);
    HUPR_LAUNCH_OK("hupr_k_colstats2");
    HUPR_LAUNCH(hupr_k_bn_finalize_bwd2, dim3((C + kFinalizeCh - 1) / kFinalizeCh), dim3(256), 0, s, partial, nblk, C, gamma1,
                       invstd1, gamma2, invstd2, 1.0f / (float)M, train, dgamma1, dbeta1, dgamma2, dbeta2, coef);
    HUPR_LAUNCH_OK("hupr_k_bn_finalize_bwd2");
    const long nv = M * C / act_v<T>();
    HUPR_LAUNCH(hupr_k_bn_bwd_apply2<T>, dim3(ew_grid(nv)), dim3(256), 0, s, dy, y_mask, x1, x2, mean1, mean2, coef, dx1, dx2,
                       nv, C, fs1, ft1, fs2, ft2);
    HUPR_LAUNCH_OK("hupr_k_bn_bwd_apply2");
    return HUPR_OK;
}
extern "C" int hupr_bn_bwd2_f32(const float* dy, const float* y_mask, const float* x1, const float* mean1, const float* invstd1,
                                const float* gamma1, const float* x2, const float* mean2, const float* invstd2,
                                const float* gamma2, float* dx1, float* dx2, float* dgamma1, float* dbeta1, float* dgamma2,
                                float* dbeta2, long M, int C, int train, void* ws, size_t ws_bytes, hupr_stream_t stream) {
    return bn_bwd2("hupr_bn_bwd2_f32", dy, y_mask, nullptr, x1, mean1, invstd1, gamma1, x2, mean2, invstd2, gamma2, dx1, dx2, dgamma1,
                   dbeta1, dgamma2, dbeta2, M, C, train, ws, ws_bytes, stream);
}
extern "C" int hupr_bn_bwd2_bf16act(const void* dy, const void* y_mask, const void* x1, const float* mean1, const float* invstd1,
                                    const float* gamma1, const void* x2, const float* mean2, const float* invstd2,
                                    const float* gamma2, void* dx1, void* dx2, float* dgamma1, float* dbeta1, float* dgamma2,
                                    float* dbeta2, long M, int C, int train, void* ws, size_t ws_bytes, hupr_stream_t stream) {
    return bn_bwd2("hupr_bn_bwd2_bf16act", static_cast<const __bf16*>(dy), static_cast<const __bf16*>(y_mask), nullptr,
                   static_cast<const __bf16*>(x1), mean1, invstd1, gamma1, static_cast<const __bf16*>(x2), mean2, invstd2, gamma2,
                   static_cast<__bf16*>(dx1), static_cast<__bf16*>(dx2), dgamma1, dbeta1, dgamma2, dbeta2, M, C, train, ws,
                   ws_bytes, stream);
}

// two-branch backward with the ReLU mask recomputed from x1, x2 and the forward scale / shift pairs (no y_mask read)
extern "C" int hupr_bn_bwd2_remask_f32(const float* dy, const float* x1, const float* fwd_scale1, const float* fwd_shift1,
                                       const float* mean1, const float* invstd1, const float* gamma1, const float* x2,
                                       const float* fwd_scale2, const float* fwd_shift2, const float* mean2,
                                       const float* invstd2, const float* gamma2, float* dx1, float* dx2, float* dgamma1,
                                       float* dbeta1, float* dgamma2, float* dbeta2, long M, int C, int train, void* ws,
                                       size_t ws_bytes, hupr_stream_t stream) {
    const float* fwd[4] = {fwd_scale1, fwd_shift1, fwd_scale2, fwd_shift2};
    return bn_bwd2("hupr_bn_bwd2_remask_f32", dy, (const float*)nullptr, fwd, x1, mean1, invstd1, gamma1, x2, mean2, invstd2, gamma2,
                   dx1, dx2, dgamma1, dbeta1, dgamma2, dbeta2, M, C, train, ws, ws_bytes, stream);
}
extern "C" int hupr_bn_bwd2_remask_bf16act(const void* dy, const void* x1, const float* fwd_scale1, const float* fwd_shift1,
                                           const float* mean1, const float* invstd1, const float* gamma1, const void* x2,
                                           const float* fwd_scale2, const float* fwd_shift2, const float* mean2,
                                           const float* invstd2, const float* gamma2, void* dx1, void* dx2, float* dgamma1,
                                           float* dbeta1, float* dgamma2, float* dbeta2, long M, int C, int train, void* ws,
                                           size_t ws_bytes, hupr_stream_t stream) {
    const float* fwd[4] = {fwd_scale1, fwd_shift1, fwd_scale2, fwd_shift2};
    return bn_bwd2("hupr_bn_bwd2_remask_bf16act", static_cast<const __bf16*>(dy), (const __bf16*)nullptr, fwd,
                   static_cast<const __bf16*>(x1), mean1, invstd1, gamma1, static_cast<const __bf16*>(x2), mean2, invstd2, gamma2,
                   static_cast<__bf16*>(dx1), static_cast<__bf16*>(dx2), dgamma1, dbeta1, dgamma2, dbeta2, M, C, train, ws,
                   ws_bytes, stream);
}

template <typename T>
static int prelu_fwd(const char* who, const T* x, const float* alpha, T* y, long n, hupr_stream_t stream) {
    HUPR_REQUIRE(x && alpha && y && n > 0 && n % act_v<T>() == 0, "%s: bad argument", who);
    const long nv = n / act_v<T>();
    HUPR_LAUNCH(hupr_k_prelu_fwd<T>, dim3(ew_grid(nv)), dim3(256), 0, as_stream(stream), x, alpha, y, nv);
    HUPR_LAUNCH_OK("hupr_k_prelu_fwd");
    return HUPR_OK;
}
extern "C" int hupr_prelu_fwd_f32(const float* x, const float* alpha, float* y, long n, hupr_stream_t stream) {
    return prelu_fwd("hupr_prelu_fwd_f32", x, alpha, y, n, stream);
}
extern "C" int hupr_prelu_fwd_bf16act(const void* x, const float* alpha, void* y, long n, hupr_stream_t stream) {
    return prelu_fwd("hupr_prelu_fwd_bf16act", static_cast<const __bf16*>(x), alpha, static_cast<__bf16*>(y), n, stream);
}

extern "C" size_t hupr_prelu_ws_bytes(void) { return 4096 * sizeof(double); }

template <typename T>
static int prelu_bwd(const char* who, const T* dy, const T* x, const float* alpha, T* dx, float* dalpha, long n, void* ws,
                     size_t ws_bytes, hupr_stream_t stream, int* n_partials = nullptr) {
    // n_partials != null: dx only; the per-workgroup partial sums of the slope gradient stay in ws (*n_partials of them) for a later
    // hupr_sum_partials_multi
    HUPR_REQUIRE(dy && x && alpha && dx && (dalpha || n_partials) && ws && n > 0 && n % act_v<T>() == 0, "%s: bad argument", who);
    if (ws_bytes < hupr_prelu_ws_bytes()) return fail(HUPR_ERR_WORKSPACE, "%s: workspace too small", who);
    hipStream_t s = as_stream(stream);
    const long nv = n / act_v<T>();
    const int grid = ew_grid(nv);
    double* partial = reinterpret_cast<double*>(ws);
    HUPR_LAUNCH(hupr_k_prelu_bwd<T>, dim3(grid), dim3(256), 0, s, dy, x, alpha, dx, nv, partial);
    HUPR_LAUNCH_OK("hupr_k_prelu_bwd");
    if (n_partials) { *n_partials = grid; return HUPR_OK; }
    HUPR_LAUNCH(hupr_k_sum_partials, dim3(1), dim3(256), 0, s, partial, grid, dalpha);
    HUPR_LAUNCH_OK("hupr_k_sum_partials");
    return HUPR_OK;
}
extern "C" int hupr_prelu_bwd_f32(const float* dy, const float* x, const float* alpha, float* dx, float* dalpha,
                                  long n, void* ws, size_t ws_bytes, hupr_stream_t stream) {
    return prelu_bwd("hupr_prelu_bwd_f32", dy, x, alpha, dx, dalpha, n, ws, ws_bytes, stream);
}
extern "C" int hupr_prelu_bwd_bf16act(const void* dy, const void* x, const float* alpha, void* dx, float* dalpha, long n,
                                      void* ws, size_t ws_bytes, hupr_stream_t stream) {
    return prelu_bwd("hupr_prelu_bwd_bf16act", static_cast<const __bf16*>(dy), static_cast<const __bf16*>(x), alpha,
                     static_cast<__bf16*>(dx), dalpha, n, ws, ws_bytes, stream);
}

// The PReLU backward without its final sum: dx, and *n_partials partial sums of the slope gradient left in `partials` (>=
// hupr_prelu_ws_bytes(), the caller's own buffer: it must stay untouched until hupr_sum_partials_multi has consumed it)
extern "C" int hupr_prelu_bwd_partials_f32(const float* dy, const float* x, const float* alpha, float* dx, long n, void* partials,
                                           size_t partials_bytes, int* n_partials, hupr_stream_t stream) {
    HUPR_REQUIRE(n_partials, "hupr_prelu_bwd_partials_f32: null pointer");
    return prelu_bwd("hupr_prelu_bwd_partials_f32", dy, x, alpha, dx, nullptr, n, partials, partials_bytes, stream, n_partials);
}
extern "C" int hupr_prelu_bwd_partials_bf16act(const void* dy, const void* x, const float* alpha, void* dx, long n, void* partials,
                                               size_t partials_bytes, int* n_partials, hupr_stream_t stream) {
    HUPR_REQUIRE(n_partials, "hupr_prelu_bwd_partials_bf16act: null pointer");
    return prelu_bwd("hupr_prelu_bwd_partials_bf16act", static_cast<const __bf16*>(dy), static_cast<const __bf16*>(x), alpha,
                     static_cast<__bf16*>(dx), nullptr, n, partials, partials_bytes, stream, n_partials);
}
// out[i][0] = sum of the n[i] doubles at partial[i], i < n_items, in ceil(n_items / 16) launches; each sum as hupr_prelu_bwd_* forms it
extern "C" int hupr_sum_partials_multi(const hupr_sum_item* items, int n_items, hupr_stream_t stream) {
    HUPR_REQUIRE(items && n_items > 0, "hupr_sum_partials_multi: bad argument");
    for (int i0 = 0; i0 < n_items; i0 += 16) {
        SumItems t{};
        t.n_items = n_items - i0 < 16 ? n_items - i0 : 16;
        for (int i = 0; i < t.n_items; ++i) {
            const hupr_sum_item& it = items[i0 + i];
            HUPR_REQUIRE(it.partial && it.out && it.n > 0, "hupr_sum_partials_multi: bad item %d", i0 + i);
            t.partial[i] = static_cast<const double*>(it.partial);
            t.n[i] = it.n;
            t.out[i] = it.out;
        }
        HUPR_LAUNCH(hupr_k_sum_partials_multi, dim3(t.n_items), dim3(256), 0, as_stream(stream), t);
        HUPR_LAUNCH_OK("hupr_k_sum_partials_multi");
    }
    return HUPR_OK;
}

// out[c] = sum over rows of x[row][c]   (bias gradient of a convolution)
template <typename T>
static int colsum(const char* who, const T* x, long M, int C, float* out, void* ws, size_t ws_bytes, hupr_stream_t stream) {
    HUPR_REQUIRE(x && out && ws, "%s: null pointer", who);
    int rc = bn_check(who, M, C, act_v<T>());
    if (rc) return rc;
    if (ws_bytes < hupr_bn_ws_bytes(C)) return fail(HUPR_ERR_WORKSPACE, "%s: workspace too small", who);
    hipStream_t s = as_stream(stream);
    const int nblk = (int)min((long)kStatBlocks, (M + 63) / 64);
    double* partial = reinterpret_cast<double*>(ws);
    HUPR_LAUNCH((hupr_k_colstats<0, T>), dim3(nblk), dim3(256), stats_lds<T>(C, 2), s, x, (const T*)nullptr,
                       (const T*)nullptr, nullptr, nullptr, M, C, partial, (const float*)nullptr, (const float*)nullptr);
    HUPR_LAUNCH_OK("hupr_k_colstats<0>");
    HUPR_LAUNCH(hupr_k_colsum_final, dim3((C + kFinalizeCh - 1) / kFinalizeCh), dim3(256), 0, s, partial, nblk, C, out);
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
    HUPR_LAUNCH((hupr_k_cast<float, __bf16>), dim3(ew_grid(n / 4)), dim3(256), 0, as_stream(stream), x,
                       static_cast<__bf16*>(y), n / 4);
    HUPR_LAUNCH_OK("hupr_k_cast");
    return HUPR_OK;
}
extern "C" int hupr_cast_bf16_to_f32(const void* x, float* y, long n, hupr_stream_t stream) {
    HUPR_REQUIRE(x && y && n > 0 && n % 4 == 0, "hupr_cast_bf16_to_f32: bad argument");
    HUPR_LAUNCH((hupr_k_cast<__bf16, float>), dim3(ew_grid(n / 4)), dim3(256), 0, as_stream(stream),
                       static_cast<const __bf16*>(x), y, n / 4);
    HUPR_LAUNCH_OK("hupr_k_cast");
    return HUPR_OK;
}
