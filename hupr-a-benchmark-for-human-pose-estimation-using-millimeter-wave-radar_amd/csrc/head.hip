// Row softmax (attention), PRGCN adjacency/bias epilogue, sigmoid heads, BCE loss, Gaussian
// targets, arg-max decode and the fused Adam step.  All HBM-bound or tiny.
//
// Reference semantics: models/layers.py:126-133 (softmax over keys), models/gcn_networks.py:23-29,
// 53-64 (X.A, W.(XA)+b, ReLU, sigmoid), models/networks.py:40, misc/losses.py:23-45,
// misc/utils.py:6-66, misc/metrics.py:10-38, tools/base.py:44-47 (Adam, coupled L2 decay).
#include "hupr_common.h"

namespace hupr {

__device__ __forceinline__ float block_reduce(float v, bool is_max, float* red) {
    v = is_max ? wave_max(v) : wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
}

// in-place softmax of each row of s[rows][n] (n % 4 == 0); one 256-thread block per row
__global__ __launch_bounds__(256) void hupr_k_softmax_rows(float* __restrict__ s, int n) {
    __shared__ float red[4];
    float4* row = reinterpret_cast<float4*>(s + (long)blockIdx.x * n);
    const int n4 = n >> 2;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < n4; i += 256) {
        const float4 v = row[i];
        mx = fmaxf(mx, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
    }
    mx = block_reduce(mx, true, red);
    float sum = 0.f;
    for (int i = threadIdx.x; i < n4; i += 256) {
        float4 v = row[i];
        v.x = expf(v.x - mx); v.y = expf(v.y - mx); v.z = expf(v.z - mx); v.w = expf(v.w - mx);
        sum += (v.x + v.y) + (v.z + v.w);
        row[i] = v;
    }
    sum = block_reduce(sum, false, red);
    const float inv = 1.f / sum;
    for (int i = threadIdx.x; i < n4; i += 256) {
        float4 v = row[i];
        v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
        row[i] = v;
    }
}

// in place on dp: ds = p * (dp - sum_j dp_j p_j)
__global__ __launch_bounds__(256) void hupr_k_softmax_rows_bwd(const float* __restrict__ p, float* __restrict__ dp, int n) {
    __shared__ float red[4];
    const float4* pr = reinterpret_cast<const float4*>(p + (long)blockIdx.x * n);
    float4* gr = reinterpret_cast<float4*>(dp + (long)blockIdx.x * n);
    const int n4 = n >> 2;
    float dot = 0.f;
    for (int i = threadIdx.x; i < n4; i += 256) {
        const float4 a = pr[i], g = gr[i];
        dot += (a.x * g.x + a.y * g.y) + (a.z * g.z + a.w * g.w);
    }
    dot = block_reduce(dot, false, red);
    for (int i = threadIdx.x; i < n4; i += 256) {
        const float4 a = pr[i];
        float4 g = gr[i];
        g.x = a.x * (g.x - dot); g.y = a.y * (g.y - dot); g.z = a.z * (g.z - dot); g.w = a.w * (g.w - dot);
        gr[i] = g;
    }
}

// ---- PRGCN epilogue: y[b][f][k'] = act( sum_k t[b][f][k] A[k][k'] + bias[f][k'] ) --------------
// t, y have row stride ld (>= K), pad columns are written as 0
// One thread per (row, output key-point): the 16-float row is read as four float4 by every lane of its 16-lane group (same
// address: broadcast), the adjacency column comes from LDS.  (The first version gave a thread the whole row in arrays indexed by
// the RUN-TIME K — private arrays in scratch memory: 14.6 us for 64 KB of data, three times per single-sample forward.)
__global__ __launch_bounds__(256) void hupr_k_gcn_adj_fwd(const float* __restrict__ t, const float* __restrict__ adj,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          long rows, int F, int K, int ld, int relu, int slices) {
    // slices > 1 (single-sample inference): t is [slices][rows][ld], the K slices of the product W x — summed here, in slice order
    __shared__ float sa[16 * 16];
    for (int i = threadIdx.x; i < 256; i += 256) sa[i] = (i / 16 < K && i % 16 < K) ? adj[(i / 16) * K + (i % 16)] : 0.f;
    __syncthreads();
    const int kp = threadIdx.x & 15;
    for (long r = (long)blockIdx.x * 16 + (threadIdx.x >> 4); r < rows; r += (long)gridDim.x * 16) {
        const int f = r % F;
        float s = kp < K ? bias[f * K + kp] : 0.f;
        if (ld == 16) {
            const float4* row = reinterpret_cast<const float4*>(t + r * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v = row[q];
                for (int sl = 1; sl < slices; ++sl) {
                    const float4 u = row[(long)sl * rows * 4 + q];
                    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
                }
                s = fmaf(v.x, sa[(4 * q) * 16 + kp], s);
                s = fmaf(v.y, sa[(4 * q + 1) * 16 + kp], s);
                s = fmaf(v.z, sa[(4 * q + 2) * 16 + kp], s);
                s = fmaf(v.w, sa[(4 * q + 3) * 16 + kp], s);
            }
        } else {
            for (int k = 0; k < K; ++k) {
                float v = t[r * ld + k];
                for (int sl = 1; sl < slices; ++sl) v += t[((long)sl * rows + r) * ld + k];
                s = fmaf(v, sa[k * 16 + kp], s);
            }
        }
        if (kp < ld) y[r * ld + kp] = kp < K ? (relu ? fmaxf(s, 0.f) : s) : 0.f;
    }
}

// g = dy * [y>0] (if relu);  dt[b][f][k] = sum_k' g[k'] A[k][k'];  dbias handled by hupr_k_gcn_dbias
__global__ __launch_bounds__(256) void hupr_k_gcn_adj_bwd(const float* __restrict__ dy, const float* __restrict__ y,
                                                          const float* __restrict__ adj, float* __restrict__ dt,
                                                          float* __restrict__ gmasked, long rows, int K, int ld, int relu) {
    __shared__ float sa[16 * 16];
    for (int i = threadIdx.x; i < K * K; i += 256) sa[i] = adj[i];
    __syncthreads();
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long)gridDim.x * 256) {
        float g[16];
        for (int k = 0; k < K; ++k) {
            float v = dy[r * ld + k];
            if (relu && !(y[r * ld + k] > 0.f)) v = 0.f;
            g[k] = v;
        }
        for (int k = 0; k < ld; ++k) {
            float s = 0.f;
            if (k < K)
                for (int kp = 0; kp < K; ++kp) s = fmaf(g[kp], sa[k * K + kp], s);
            dt[r * ld + k] = s;
            gmasked[r * ld + k] = (k < K) ? g[k] : 0.f;
        }
    }
}

// dbias[f][k] = sum_b g[b][f][k]
__global__ void hupr_k_gcn_dbias(const float* __restrict__ g, float* __restrict__ dbias, int Bn, int F, int K, int ld) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F * K) return;
    const int f = i / K, k = i % K;
    float s = 0.f;
    for (int b = 0; b < Bn; ++b) s += g[((long)b * F + f) * ld + k];
    dbias[i] = s;
}

// ---- heads: x (B, HW, ld) channels-last logits -> y (B, K, HW) probabilities (NCHW) --------------
__global__ void hupr_k_sigmoid_to_nchw(const float* __restrict__ x, float* __restrict__ y, int Bn, int HW, int K, int ld) {
    const long total = (long)Bn * K * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int hw = i % HW;
        const long t = i / HW;
        const int k = t % K, b = t / K;
        const float v = x[((long)b * HW + hw) * ld + k];
        y[i] = 1.f / (1.f + expf(-v));
    }
}
// dx (B,HW,ld) = dy (B,K,HW) * y (1-y), pad channels zero
__global__ void hupr_k_sigmoid_to_nchw_bwd(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx,
                                           int Bn, int HW, int K, int ld) {
    const long total = (long)Bn * HW * ld;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = i % ld;
        const long t = i / ld;
        const int hw = t % HW, b = t / HW;
        float v = 0.f;
        if (k < K) {
            const long j = ((long)b * K + k) * HW + hw;
            const float p = y[j];
            v = dy[j] * p * (1.f - p);
        }
        dx[i] = v;
    }
}

// ---- BCE (nn.BCELoss, mean reduction, log clamped at -100 like PyTorch) -----------------------
__global__ __launch_bounds__(256) void hupr_k_bce_fwd(const float* __restrict__ p, const float* __restrict__ t, long n,
                                                      double* __restrict__ partial) {
    __shared__ double red[4];
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float pv = p[i], tv = t[i];
        const float l1 = fmaxf(logf(pv), -100.f), l0 = fmaxf(logf(1.f - pv), -100.f);
        acc -= tv * l1 + (1.f - tv) * l0;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = (double)acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void hupr_k_bce_final(const double* __restrict__ partial, int nblk, double inv_n, float* __restrict__ out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 64) s += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) out[0] = (float)(s * inv_n);
}
// The two BCE losses of a step (first head and PRGCN head against the same targets, reference misc/losses.py:24-33) and their
// weighted sum in two launches instead of four + three torch-native ones: blockIdx.y = head in the partial pass; the final launch forms
// both means in hupr_k_bce_final's order (one wave per head) and loss = alpha * loss1 + beta * loss2 as torch forms it (two roundings
// of the products, one of the sum: no fma).
__global__ __launch_bounds__(256) void hupr_k_bce_pair_fwd(const float* __restrict__ p1, const float* __restrict__ p2,
                                                           const float* __restrict__ t, long n, double* __restrict__ partial) {
    __shared__ double red[4];
    const float* __restrict__ p = blockIdx.y ? p2 : p1;
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float pv = p[i], tv = t[i];
        const float l1 = fmaxf(logf(pv), -100.f), l0 = fmaxf(logf(1.f - pv), -100.f);
        acc -= tv * l1 + (1.f - tv) * l0;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = (double)acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(long)blockIdx.y * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void hupr_k_bce_pair_final(const double* __restrict__ partial, int nblk, double inv_n, float alpha, float beta,
                                      float* __restrict__ out /* loss, loss1, loss2 */) {
    __shared__ float l[2];
    const int head = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double s = 0.0;
    for (int i = lane; i < nblk; i += 64) s += partial[(long)head * nblk + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) l[head] = (float)(s * inv_n);
    __syncthreads();
    if (threadIdx.x == 0) {
        out[1] = l[0];
        out[2] = l[1];
        out[0] = __fadd_rn(__fmul_rn(alpha, l[0]), __fmul_rn(beta, l[1]));
    }
}
// dp1 = (g alpha / n) (p1 - t) / max(p1 (1 - p1), 1e-12), dp2 likewise with (g beta [+ g2]) / n
__global__ void hupr_k_bce_pair_bwd(const float* __restrict__ p1, const float* __restrict__ p2, const float* __restrict__ t,
                                    const float* __restrict__ g, const float* __restrict__ g2, float alpha, float beta, float inv_n,
                                    float* __restrict__ dp1, float* __restrict__ dp2, long n) {
    const bool second = blockIdx.y != 0;
    const float* __restrict__ p = second ? p2 : p1;
    float* __restrict__ dp = second ? dp2 : dp1;
    float go = __fmul_rn(g[0], second ? beta : alpha);
    if (second && g2) go = __fadd_rn(go, g2[0]);
    const float gs = go * inv_n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float pv = p[i];
        dp[i] = gs * (pv - t[i]) / fmaxf(pv * (1.f - pv), 1e-12f);
    }
}
// dp = gscale * (p - t) / max(p (1-p), 1e-12)     (PyTorch binary_cross_entropy_backward)
__global__ void hupr_k_bce_bwd(const float* __restrict__ p, const float* __restrict__ t, const float* __restrict__ gout,
                               float inv_n, float* __restrict__ dp, long n) {
    const float gs = gout[0] * inv_n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float pv = p[i];
        dp[i] = gs * (pv - t[i]) / fmaxf(pv * (1.f - pv), 1e-12f);
    }
}

// ---- Gaussian targets: joints (B,K,2) int64 image px -> t (B,K,H,W); patch = host table (2*rad+1)^2 --
__global__ void hupr_k_gaussian_targets(const long long* __restrict__ joints, const float* __restrict__ patch,
                                        float* __restrict__ t, int BK, int H, int rad, float stride) {
    const long total = (long)BK * H * H;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = i % H;
        const long q = i / H;
        const int y = q % H, bk = q / H;
        const int mx = (int)((float)joints[bk * 2 + 0] / stride + 0.5f);
        const int my = (int)((float)joints[bk * 2 + 1] / stride + 0.5f);
        float v = 0.f;
        const bool outside = (mx - rad >= H) || (my - rad >= H) || (mx + rad + 1 < 0) || (my + rad + 1 < 0);
        const int dx = x - mx + rad, dy = y - my + rad, sz = 2 * rad + 1;
        if (!outside && dx >= 0 && dx < sz && dy >= 0 && dy < sz) v = patch[dy * sz + dx];
        t[i] = v;
    }
}

// ---- arg-max over HW per (b,k) row, first maximum wins (np.argmax) --------------------------------
__global__ __launch_bounds__(64) void hupr_k_argmax_rows(const float* __restrict__ p, int n, int* __restrict__ idx,
                                                         float* __restrict__ maxval) {
    const float* row = p + (long)blockIdx.x * n;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += 64) {
        const float v = row[i];
        if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (threadIdx.x == 0) { idx[blockIdx.x] = bi; maxval[blockIdx.x] = best; }
}

// ---- Adam with coupled L2 weight decay (torch.optim.Adam semantics), one flat launch --------------
__global__ __launch_bounds__(256) void hupr_k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                   float wd, float bc1, float bc2_sqrt, float gscale) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float pv = p[i];
        const float gr = fmaf(wd, pv, g[i] * gscale);
        const float mv = fmaf(b1, m[i], (1.f - b1) * gr);
        const float vv = fmaf(b2, v[i], (1.f - b2) * gr * gr);
        m[i] = mv;
        v[i] = vv;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        p[i] = pv - (lr / bc1) * (mv / denom);
    }
}

// same update with the learning rate and the step count read from device memory (state = {lr, step}): the launch
// arguments of a captured hipGraph are frozen, the bias corrections must not be
__global__ __launch_bounds__(256) void hupr_k_adam_dev(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, long n, const float* __restrict__ state, float b1,
                                                       float b2, float eps, float wd, float gscale) {
    const float lr = state[0];
    const double step = (double)state[1];
    const float bc1 = (float)(1.0 - pow((double)b1, step)), bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, step));
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float pv = p[i];
        const float gr = fmaf(wd, pv, g[i] * gscale);
        const float mv = fmaf(b1, m[i], (1.f - b1) * gr);
        const float vv = fmaf(b2, v[i], (1.f - b2) * gr * gr);
        m[i] = mv;
        v[i] = vv;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        p[i] = pv - (lr / bc1) * (mv / denom);
    }
}

static inline int grid1d(long n, int bs = 256, long cap = 4096) { return (int)min(cap, (n + bs - 1) / bs); }


// ------------------------------------------------------------------------------------------
// 1x1 key-point head (reference models/layers.py:94, nn.Conv2d(32, 14, 1, bias=False)) in plain fp32 FMAs.
// 58 MFLOP per batch of 32: the generic implicit-GEMM path spends its time on tile setup, zero-padded K axes and split-K
// partials (0.26 ms per training step when the "head" region of a bf16 run is switched to fp32, functional.PRECISION).
//   x [M][32] fp32, w [16][32] (rows >= 14 zero), y / dy [M][16], dx [M][32], dw [16][32]
// One thread per voxel (its 32 input channels in registers, the weights through scalar loads: the index is uniform).
// ------------------------------------------------------------------------------------------
constexpr int kHeadCi = 32, kHeadCo = 16;

__global__ __launch_bounds__(256) void hupr_k_head1x1_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                          float* __restrict__ y, long M) {
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    if (v >= M) return;
    float xr[kHeadCi];
    const float4* xp = reinterpret_cast<const float4*>(x + v * kHeadCi);
#pragma unroll
    for (int i = 0; i < kHeadCi / 4; ++i) { const float4 t = xp[i]; xr[4 * i] = t.x; xr[4 * i + 1] = t.y; xr[4 * i + 2] = t.z; xr[4 * i + 3] = t.w; }
    float4* yp = reinterpret_cast<float4*>(y + v * kHeadCo);
#pragma unroll
    for (int k4 = 0; k4 < kHeadCo / 4; ++k4) {
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < kHeadCi; ++c) a = fmaf(xr[c], w[(4 * k4 + j) * kHeadCi + c], a);
            o[j] = a;
        }
        yp[k4] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// dx[v][c] = sum_k dy[v][k] w[k][c]
__global__ __launch_bounds__(256) void hupr_k_head1x1_dgrad(const float* __restrict__ dy, const float* __restrict__ w,
                                                            float* __restrict__ dx, long M) {
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    if (v >= M) return;
    float g[kHeadCo];
    const float4* gp = reinterpret_cast<const float4*>(dy + v * kHeadCo);
#pragma unroll
    for (int i = 0; i < kHeadCo / 4; ++i) { const float4 t = gp[i]; g[4 * i] = t.x; g[4 * i + 1] = t.y; g[4 * i + 2] = t.z; g[4 * i + 3] = t.w; }
    float4* xp = reinterpret_cast<float4*>(dx + v * kHeadCi);
#pragma unroll
    for (int c4 = 0; c4 < kHeadCi / 4; ++c4) {
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < kHeadCo; ++k) a = fmaf(g[k], w[k * kHeadCi + 4 * c4 + j], a);
            o[j] = a;
        }
        xp[c4] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// dw[k][c] = sum_v dy[v][k] x[v][c]: workgroup b sums voxels [b * per, (b + 1) * per) in chunks of 128 staged in LDS; thread
// t owns outputs (k, c) = (t >> 4, 2 (t & 15) + {0, 1}); partial rows [grid][512] are summed in block order by the second
// kernel (deterministic, no atomics).
constexpr int kHeadWgradGrid = 512;      // partial rows (a workgroup's slice is a serial load -> barrier -> multiply chain: keep it short)
__global__ __launch_bounds__(256) void hupr_k_head1x1_wgrad(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ part, long M) {
    __shared__ float xs[128][kHeadCi + 1];
    __shared__ float gs[128][kHeadCo + 1];
    const int tid = threadIdx.x, k = tid >> 4, c0 = 2 * (tid & 15);
    const long per = (M + gridDim.x - 1) / gridDim.x, v0 = (long)blockIdx.x * per, v1 = min(M, v0 + per);
    float a0 = 0.f, a1 = 0.f;
    for (long vb = v0; vb < v1; vb += 128) {
        const int n = (int)min((long)128, v1 - vb);
        __syncthreads();
        for (int i = tid; i < 128 * kHeadCi; i += 256) { const int r = i >> 5, c = i & 31; xs[r][c] = r < n ? x[(vb + r) * kHeadCi + c] : 0.f; }
        for (int i = tid; i < 128 * kHeadCo; i += 256) { const int r = i >> 4, c = i & 15; gs[r][c] = r < n ? dy[(vb + r) * kHeadCo + c] : 0.f; }
        __syncthreads();
#pragma unroll 8
        for (int r = 0; r < 128; ++r) {
            const float g = gs[r][k];
            a0 = fmaf(g, xs[r][c0], a0);
            a1 = fmaf(g, xs[r][c0 + 1], a1);
        }
    }
    part[(long)blockIdx.x * 512 + k * kHeadCi + c0] = a0;
    part[(long)blockIdx.x * 512 + k * kHeadCi + c0 + 1] = a1;
}
// dw[i] = sum over the partial rows, in a fixed order: 8 workgroups x (64 outputs x 4 row groups), 16 loads in flight per thread
__global__ __launch_bounds__(256) void hupr_k_head1x1_wgrad_reduce(const float* __restrict__ part, float* __restrict__ dw, int rows, int n_out) {
    __shared__ double red[4][64];
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    double a = 0.0;
#pragma unroll 16
    for (int r = g; r < rows; r += 4) a += (double)part[(long)r * 512 + i];
    red[g][threadIdx.x & 63] = a;
    __syncthreads();
    if (g == 0 && i < n_out) dw[i] = (float)((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

}  // namespace hupr

using namespace hupr;

extern "C" int hupr_softmax_rows_f32(float* s, long rows, int n, hupr_stream_t stream) {
    HUPR_REQUIRE(s && rows > 0 && n > 0 && n % 4 == 0 && rows < (1L << 31), "hupr_softmax_rows_f32: bad argument");
    HUPR_LAUNCH(hupr_k_softmax_rows, dim3((unsigned)rows), dim3(256), 0, as_stream(stream), s, n);
    HUPR_LAUNCH_OK("hupr_k_softmax_rows");
    return HUPR_OK;
}
extern "C" int hupr_softmax_rows_bwd_f32(const float* p, float* dp_inout, long rows, int n, hupr_stream_t stream) {
    HUPR_REQUIRE(p && dp_inout && rows > 0 && n > 0 && n % 4 == 0 && rows < (1L << 31), "hupr_softmax_rows_bwd_f32: bad argument");
    HUPR_LAUNCH(hupr_k_softmax_rows_bwd, dim3((unsigned)rows), dim3(256), 0, as_stream(stream), p, dp_inout, n);
    HUPR_LAUNCH_OK("hupr_k_softmax_rows_bwd");
    return HUPR_OK;
}

extern "C" size_t hupr_head1x1_ws_bytes(void) { return (size_t)kHeadWgradGrid * 512 * sizeof(float); }
extern "C" int hupr_head1x1_fwd_f32(const float* x, const float* w16, float* y, long M, hupr_stream_t stream) {
    HUPR_REQUIRE(M >= 0, "hupr_head1x1_fwd_f32: M=%ld", M);
    if (M == 0) return HUPR_OK;
    HUPR_REQUIRE(x && w16 && y && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "hupr_head1x1_fwd_f32: null or misaligned pointer");
    HUPR_LAUNCH(hupr_k_head1x1_fwd, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, as_stream(stream), x, w16, y, M);
    HUPR_LAUNCH_OK("hupr_k_head1x1_fwd");
    return HUPR_OK;
}
// out_rows: how many of the 16 filter rows of dw are written (16: the padded layout; K = 14: the parameter itself, e.g. its slot in a
// flat gradient bucket, whose neighbours must not be touched)
extern "C" int hupr_head1x1_bwd_rows_f32(const float* x, const float* w16, const float* dy, float* dx_or_null, float* dw16_or_null,
                                         int out_rows, long M, void* ws, size_t ws_bytes, hupr_stream_t stream) {
    HUPR_REQUIRE(M >= 0 && out_rows >= 1 && out_rows <= 16, "hupr_head1x1_bwd_f32: M=%ld out_rows=%d", M, out_rows);
    if (M == 0) return HUPR_OK;
    HUPR_REQUIRE(x && w16 && dy && ((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)dx_or_null & 15) == 0,
                 "hupr_head1x1_bwd_f32: null or misaligned pointer");
    if (dx_or_null) {
        HUPR_LAUNCH(hupr_k_head1x1_dgrad, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, as_stream(stream), dy, w16, dx_or_null, M);
        HUPR_LAUNCH_OK("hupr_k_head1x1_dgrad");
    }
    if (dw16_or_null) {
        if (!ws || ws_bytes < hupr_head1x1_ws_bytes()) return fail(HUPR_ERR_WORKSPACE, "hupr_head1x1_bwd_f32: workspace %zu < %zu", ws_bytes, hupr_head1x1_ws_bytes());
        const int grid = (int)min((long)kHeadWgradGrid, (M + 127) / 128);
        HUPR_LAUNCH(hupr_k_head1x1_wgrad, dim3(grid), dim3(256), 0, as_stream(stream), x, dy, static_cast<float*>(ws), M);
        HUPR_LAUNCH(hupr_k_head1x1_wgrad_reduce, dim3(8), dim3(256), 0, as_stream(stream), static_cast<const float*>(ws), dw16_or_null, grid,
                    out_rows * 32);
        HUPR_LAUNCH_OK("hupr_k_head1x1_wgrad");
    }
    return HUPR_OK;
}
extern "C" int hupr_head1x1_bwd_f32(const float* x, const float* w16, const float* dy, float* dx_or_null, float* dw16_or_null,
                                    long M, void* ws, size_t ws_bytes, hupr_stream_t stream) {
    return hupr_head1x1_bwd_rows_f32(x, w16, dy, dx_or_null, dw16_or_null, 16, M, ws, ws_bytes, stream);
}

extern "C" int hupr_gcn_adj_fwd_f32(const float* t, const float* adj, const float* bias, float* y, int Bn, int F, int K,
                                    int ld, int relu, hupr_stream_t stream) {
    HUPR_REQUIRE(t && adj && bias && y && Bn > 0 && F > 0 && K > 0 && K <= 16 && ld >= K && ld <= 16, "hupr_gcn_adj_fwd_f32: bad argument");
    const long rows = (long)Bn * F;
    HUPR_LAUNCH(hupr_k_gcn_adj_fwd, dim3((unsigned)min((long)4096, (rows + 15) / 16)), dim3(256), 0, as_stream(stream), t, adj, bias, y, rows, F, K, ld, relu, 1);
    HUPR_LAUNCH_OK("hupr_k_gcn_adj_fwd");
    return HUPR_OK;
}
// the same with t given as `slices` partial products [slices][Bn * F][ld] (K slices of W x: single-sample inference), summed in slice order
extern "C" int hupr_gcn_adj_fwd_sliced_f32(const float* t, int slices, const float* adj, const float* bias, float* y, int Bn, int F,
                                           int K, int ld, int relu, hupr_stream_t stream) {
    HUPR_REQUIRE(t && adj && bias && y && Bn > 0 && F > 0 && K > 0 && K <= 16 && ld >= K && ld <= 16 && slices >= 1 && slices <= 64,
                 "hupr_gcn_adj_fwd_sliced_f32: bad argument");
    const long rows = (long)Bn * F;
    HUPR_LAUNCH(hupr_k_gcn_adj_fwd, dim3((unsigned)min((long)4096, (rows + 15) / 16)), dim3(256), 0, as_stream(stream), t, adj, bias, y, rows, F, K, ld, relu, slices);
    HUPR_LAUNCH_OK("hupr_k_gcn_adj_fwd");
    return HUPR_OK;
}
extern "C" int hupr_gcn_adj_bwd_f32(const float* dy, const float* y, const float* adj, float* dt, float* gmasked,
                                    float* dbias, int Bn, int F, int K, int ld, int relu, hupr_stream_t stream) {
    HUPR_REQUIRE(dy && y && adj && dt && gmasked && dbias && Bn > 0 && F > 0 && K > 0 && K <= 16 && ld >= K && ld <= 16,
                 "hupr_gcn_adj_bwd_f32: bad argument");
    const long rows = (long)Bn * F;
    hipStream_t s = as_stream(stream);
    HUPR_LAUNCH(hupr_k_gcn_adj_bwd, dim3(grid1d(rows)), dim3(256), 0, s, dy, y, adj, dt, gmasked, rows, K, ld, relu);
    HUPR_LAUNCH_OK("hupr_k_gcn_adj_bwd");
    HUPR_LAUNCH(hupr_k_gcn_dbias, dim3((F * K + 255) / 256), dim3(256), 0, s, gmasked, dbias, Bn, F, K, ld);
    HUPR_LAUNCH_OK("hupr_k_gcn_dbias");
    return HUPR_OK;
}

extern "C" int hupr_sigmoid_to_nchw_f32(const float* x, float* y, int Bn, int HW, int K, int ld, hupr_stream_t stream) {
    HUPR_REQUIRE(x && y && Bn > 0 && HW > 0 && K > 0 && ld >= K, "hupr_sigmoid_to_nchw_f32: bad argument");
    HUPR_LAUNCH(hupr_k_sigmoid_to_nchw, dim3(grid1d((long)Bn * K * HW)), dim3(256), 0, as_stream(stream), x, y, Bn, HW, K, ld);
    HUPR_LAUNCH_OK("hupr_k_sigmoid_to_nchw");
    return HUPR_OK;
}
extern "C" int hupr_sigmoid_to_nchw_bwd_f32(const float* dy, const float* y, float* dx, int Bn, int HW, int K, int ld,
                                            hupr_stream_t stream) {
    HUPR_REQUIRE(dy && y && dx && Bn > 0 && HW > 0 && K > 0 && ld >= K, "hupr_sigmoid_to_nchw_bwd_f32: bad argument");
    HUPR_LAUNCH(hupr_k_sigmoid_to_nchw_bwd, dim3(grid1d((long)Bn * HW * ld)), dim3(256), 0, as_stream(stream), dy, y, dx, Bn, HW, K, ld);
    HUPR_LAUNCH_OK("hupr_k_sigmoid_to_nchw_bwd");
    return HUPR_OK;
}

extern "C" size_t hupr_bce_ws_bytes(void) { return 1024 * sizeof(double); }
extern "C" int hupr_bce_fwd_f32(const float* p, const float* t, long n, float* loss, void* ws, size_t ws_bytes,
                                hupr_stream_t stream) {
    HUPR_REQUIRE(p && t && loss && ws && n > 0, "hupr_bce_fwd_f32: bad argument");
    if (ws_bytes < hupr_bce_ws_bytes()) return fail(HUPR_ERR_WORKSPACE, "hupr_bce_fwd_f32: workspace too small");
    hipStream_t s = as_stream(stream);
    const int nblk = grid1d(n, 256, 1024);
    HUPR_LAUNCH(hupr_k_bce_fwd, dim3(nblk), dim3(256), 0, s, p, t, n, reinterpret_cast<double*>(ws));
    HUPR_LAUNCH_OK("hupr_k_bce_fwd");
    HUPR_LAUNCH(hupr_k_bce_final, dim3(1), dim3(64), 0, s, reinterpret_cast<const double*>(ws), nblk, 1.0 / (double)n, loss);
    HUPR_LAUNCH_OK("hupr_k_bce_final");
    return HUPR_OK;
}
// loss3 = {alpha * BCE(p1, t) + beta * BCE(p2, t), BCE(p1, t), BCE(p2, t)}; ws: 2 x hupr_bce_ws_bytes()
extern "C" int hupr_bce_pair_fwd_f32(const float* p1, const float* p2, const float* t, long n, float alpha, float beta, float* loss3,
                                     void* ws, size_t ws_bytes, hupr_stream_t stream) {
    HUPR_REQUIRE(p1 && p2 && t && loss3 && ws && n > 0, "hupr_bce_pair_fwd_f32: bad argument");
    if (ws_bytes < 2 * hupr_bce_ws_bytes()) return fail(HUPR_ERR_WORKSPACE, "hupr_bce_pair_fwd_f32: workspace too small");
    hipStream_t s = as_stream(stream);
    const int nblk = grid1d(n, 256, 1024);
    HUPR_LAUNCH(hupr_k_bce_pair_fwd, dim3(nblk, 2), dim3(256), 0, s, p1, p2, t, n, reinterpret_cast<double*>(ws));
    HUPR_LAUNCH_OK("hupr_k_bce_pair_fwd");
    HUPR_LAUNCH(hupr_k_bce_pair_final, dim3(1), dim3(128), 0, s, reinterpret_cast<const double*>(ws), nblk, 1.0 / (double)n, alpha, beta, loss3);
    HUPR_LAUNCH_OK("hupr_k_bce_pair_final");
    return HUPR_OK;
}
extern "C" int hupr_bce_pair_bwd_f32(const float* p1, const float* p2, const float* t, const float* grad_loss,
                                     const float* grad_loss2_or_null, float alpha, float beta, float* dp1, float* dp2, long n,
                                     hupr_stream_t stream) {
    HUPR_REQUIRE(p1 && p2 && t && grad_loss && dp1 && dp2 && n > 0, "hupr_bce_pair_bwd_f32: bad argument");
    HUPR_LAUNCH(hupr_k_bce_pair_bwd, dim3(grid1d(n), 2), dim3(256), 0, as_stream(stream), p1, p2, t, grad_loss, grad_loss2_or_null, alpha,
                beta, 1.0f / (float)n, dp1, dp2, n);
    HUPR_LAUNCH_OK("hupr_k_bce_pair_bwd");
    return HUPR_OK;
}
extern "C" int hupr_bce_bwd_f32(const float* p, const float* t, const float* grad_out, float* dp, long n, hupr_stream_t stream) {
    HUPR_REQUIRE(p && t && grad_out && dp && n > 0, "hupr_bce_bwd_f32: bad argument");
    HUPR_LAUNCH(hupr_k_bce_bwd, dim3(grid1d(n)), dim3(256), 0, as_stream(stream), p, t, grad_out, 1.0f / (float)n, dp, n);
    HUPR_LAUNCH_OK("hupr_k_bce_bwd");
    return HUPR_OK;
}

extern "C" int hupr_gaussian_targets_f32(const long long* joints, const float* patch, float* t, int BK, int H, int rad,
                                         float stride, hupr_stream_t stream) {
    HUPR_REQUIRE(joints && patch && t && BK > 0 && H > 0 && rad > 0 && stride > 0.f, "hupr_gaussian_targets_f32: bad argument");
    HUPR_LAUNCH(hupr_k_gaussian_targets, dim3(grid1d((long)BK * H * H)), dim3(256), 0, as_stream(stream), joints, patch, t, BK, H, rad, stride);
    HUPR_LAUNCH_OK("hupr_k_gaussian_targets");
    return HUPR_OK;
}

extern "C" int hupr_argmax_rows_f32(const float* p, long rows, int n, int* idx, float* maxval, hupr_stream_t stream) {
    HUPR_REQUIRE(p && idx && maxval && rows > 0 && n > 0 && rows < (1L << 31), "hupr_argmax_rows_f32: bad argument");
    HUPR_LAUNCH(hupr_k_argmax_rows, dim3((unsigned)rows), dim3(64), 0, as_stream(stream), p, n, idx, maxval);
    HUPR_LAUNCH_OK("hupr_k_argmax_rows");
    return HUPR_OK;
}

// step = 1-based step count after increment; gscale multiplies the gradient (e.g. 1/world_size)
extern "C" int hupr_adam_step_f32(float* p, const float* g, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                                  float beta2, float eps, float weight_decay, int step, float gscale, hupr_stream_t stream) {
    HUPR_REQUIRE(p && g && exp_avg && exp_avg_sq && n > 0 && step >= 1, "hupr_adam_step_f32: bad argument");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    HUPR_LAUNCH(hupr_k_adam, dim3(grid1d(n, 256, 8192)), dim3(256), 0, as_stream(stream), p, g, exp_avg, exp_avg_sq, n, lr,
                       beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), gscale);
    HUPR_LAUNCH_OK("hupr_k_adam");
    return HUPR_OK;
}

// Same as hupr_adam_step_f32 with {lr, step} in device memory (dev_state[0] = learning rate, dev_state[1] = step count,
// both float): usable inside a captured hipGraph whose launch arguments are frozen.
extern "C" int hupr_adam_step_dev_f32(float* p, const float* g, float* exp_avg, float* exp_avg_sq, long n,
                                      const float* dev_state, float beta1, float beta2, float eps, float weight_decay,
                                      float gscale, hupr_stream_t stream) {
    HUPR_REQUIRE(p && g && exp_avg && exp_avg_sq && dev_state && n > 0, "hupr_adam_step_dev_f32: bad argument");
    HUPR_LAUNCH(hupr_k_adam_dev, dim3(grid1d(n, 256, 8192)), dim3(256), 0, as_stream(stream), p, g, exp_avg, exp_avg_sq, n,
                       dev_state, beta1, beta2, eps, weight_decay, gscale);
    HUPR_LAUNCH_OK("hupr_k_adam_dev");
    return HUPR_OK;
}
