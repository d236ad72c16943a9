// Memory-bound spatial kernels on channels-last fp32 activations:
//   * MNet front end (elevation mean + the reference's .view chirp reinterpretation +
//     Conv3d(2->32,(2,1,1),s(2,1,1)) + MaxPool3d((4,1,1))) fused into one pass over the
//     (B,G,F,2,R,A,E) input      [models/networks.py:23-33, models/chirp_networks.py:11-21]
//   * align_corners=True linear resampling (tri-/bilinear, down or up) fwd + bwd
//     [nn.Upsample / F.interpolate call sites: models/layers.py:84,89,199,204; gcn_networks.py:49,63]
#include "hupr_common.h"

namespace hupr {

// ------------------------------------------------------------------------------------------
// MNet.  x[bg][f=8][c=2][R*A pixels][E=8] ; the reference views the 16 (f,c) planes as
// (ch2 = j/8, t = j%8) with j = 2f+c  (a memory reinterpretation, not a permute).
// conv: o[co][t2] = bias[co] + sum_{ch2,kt} W[co][ch2][kt] * v[ch2][2*t2+kt],  t2 = 0..3
// out[bg][pixel][co] = max_t2 o[co][t2]                      (channels-last, depth axis = g)
// ------------------------------------------------------------------------------------------
constexpr int kNF = 32;

__device__ __forceinline__ void mnet_load_means(const float* __restrict__ x, long plane_stride, long pix, float* m) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float4* p = reinterpret_cast<const float4*>(x + j * plane_stride + pix * 8);
        const float4 a = p[0], b = p[1];
        // same association as torch.mean's pairwise-ish sum is not reproducible; plain tree here
        m[j] = (((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w))) * 0.125f;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void hupr_k_mnet_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, T* __restrict__ out,
                                                       float* __restrict__ means, long n_bg, int pixels) {
    __shared__ float sw[kNF * 4 + kNF];
    for (int i = threadIdx.x; i < kNF * 4 + kNF; i += 256) sw[i] = (i < kNF * 4) ? w[i] : bias[i - kNF * 4];
    __syncthreads();
    const long total = n_bg * pixels;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long bg = idx / pixels, pix = idx - bg * pixels;
        const float* xb = x + bg * 16 * (long)pixels * 8;
        float m[16];
        mnet_load_means(xb, (long)pixels * 8, pix, m);
        if (means) {                                   // 16 elevation means per pixel: all the backward pass needs of x (1/8 of its bytes)
#pragma unroll
            for (int j = 0; j < 16; j += 4)
                *reinterpret_cast<float4*>(means + idx * 16 + j) = make_float4(m[j], m[j + 1], m[j + 2], m[j + 3]);
        }
        T* o = out + idx * kNF;
#pragma unroll
        for (int c4 = 0; c4 < kNF / 4; ++c4) {
            float r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int co = c4 * 4 + k;
                const float w00 = sw[co * 4 + 0], w01 = sw[co * 4 + 1], w10 = sw[co * 4 + 2], w11 = sw[co * 4 + 3];
                float best = -INFINITY;
#pragma unroll
                for (int t2 = 0; t2 < 4; ++t2) {
                    float v = sw[kNF * 4 + co];
                    v = fmaf(w00, m[2 * t2], v);
                    v = fmaf(w01, m[2 * t2 + 1], v);
                    v = fmaf(w10, m[8 + 2 * t2], v);
                    v = fmaf(w11, m[8 + 2 * t2 + 1], v);
                    best = fmaxf(best, v);
                }
                r[k] = best;
            }
            st_act4(o + c4 * 4, make_float4(r[0], r[1], r[2], r[3]));
        }
    }
}

// The same front end fed by the fused loader's elevation-mean planes mp[bg][16][pixels] (hupr_fft_chain_loader_means_f32):
// sixteen coalesced 4-byte reads per pixel instead of 512 bytes; also leaves the pixel-major means the backward pass reads.
template <typename T>
__global__ __launch_bounds__(256) void hupr_k_mnet_fwd_means(const float* __restrict__ mp, const float* __restrict__ w,
                                                             const float* __restrict__ bias, T* __restrict__ out,
                                                             float* __restrict__ means, long n_bg, int pixels) {
    __shared__ float sw[kNF * 4 + kNF];
    for (int i = threadIdx.x; i < kNF * 4 + kNF; i += 256) sw[i] = (i < kNF * 4) ? w[i] : bias[i - kNF * 4];
    __syncthreads();
    const long total = n_bg * pixels;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long bg = idx / pixels, pix = idx - bg * pixels;
        const float* xb = mp + bg * 16 * (long)pixels + pix;
        float m[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) m[j] = xb[(long)j * pixels];
        if (means) {
#pragma unroll
            for (int j = 0; j < 16; j += 4)
                *reinterpret_cast<float4*>(means + idx * 16 + j) = make_float4(m[j], m[j + 1], m[j + 2], m[j + 3]);
        }
        T* o = out + idx * kNF;
#pragma unroll
        for (int c4 = 0; c4 < kNF / 4; ++c4) {
            float r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int co = c4 * 4 + k;
                const float w00 = sw[co * 4 + 0], w01 = sw[co * 4 + 1], w10 = sw[co * 4 + 2], w11 = sw[co * 4 + 3];
                float best = -INFINITY;
#pragma unroll
                for (int t2 = 0; t2 < 4; ++t2) {
                    float v = sw[kNF * 4 + co];
                    v = fmaf(w00, m[2 * t2], v);
                    v = fmaf(w01, m[2 * t2 + 1], v);
                    v = fmaf(w10, m[8 + 2 * t2], v);
                    v = fmaf(w11, m[8 + 2 * t2 + 1], v);
                    best = fmaxf(best, v);
                }
                r[k] = best;
            }
            st_act4(o + c4 * 4, make_float4(r[0], r[1], r[2], r[3]));
        }
    }
}

// backward: recompute the arg-max chirp step, accumulate dW[co][ch2][kt] and dbias[co]
// partial[blk][160].  A thread owns one pixel x 8 output channels (four threads per pixel): 40 accumulators instead of
// 160 keeps the kernel at full occupancy — the one-pixel-x-32-channels version ran at 0.66 TB/s on 134 MB.
template <typename T>
__global__ __launch_bounds__(256) void hupr_k_mnet_bwd(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, const T* __restrict__ dy,
                                                       const float* __restrict__ means, long n_bg, int pixels,
                                                       float* __restrict__ partial) {
    __shared__ float sw[kNF * 4 + kNF];
    __shared__ float red[4][kNF * 5];
    for (int i = threadIdx.x; i < kNF * 4 + kNF; i += 256) sw[i] = (i < kNF * 4) ? w[i] : bias[i - kNF * 4];
    for (int i = threadIdx.x; i < 4 * kNF * 5; i += 256) (&red[0][0])[i] = 0.f;
    __syncthreads();
    const int cg = threadIdx.x & 3;                   // channel group: channels 8 cg .. 8 cg + 7
    float acc[40];
#pragma unroll
    for (int i = 0; i < 40; ++i) acc[i] = 0.f;
    const long total = n_bg * pixels;
    for (long idx = (long)blockIdx.x * 64 + (threadIdx.x >> 2); idx < total; idx += (long)gridDim.x * 64) {
        float m[16];
        if (means) {
#pragma unroll
            for (int j = 0; j < 16; j += 4) {
                const float4 t = *reinterpret_cast<const float4*>(means + idx * 16 + j);
                m[j] = t.x; m[j + 1] = t.y; m[j + 2] = t.z; m[j + 3] = t.w;
            }
        } else {
            const long bg = idx / pixels, pix = idx - bg * pixels;
            mnet_load_means(x + bg * 16 * (long)pixels * 8, (long)pixels * 8, pix, m);
        }
        const T* g4 = dy + idx * kNF + cg * 8;
#pragma unroll
        for (int c4 = 0; c4 < 2; ++c4) {
            const float4 gv = ld_act4(g4 + c4 * 4);
            const float gs[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int cl = c4 * 4 + k, co = cg * 8 + cl;
                const float w00 = sw[co * 4 + 0], w01 = sw[co * 4 + 1], w10 = sw[co * 4 + 2], w11 = sw[co * 4 + 3];
                float best = -INFINITY, a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
#pragma unroll
                for (int t2 = 0; t2 < 4; ++t2) {
                    float v = sw[kNF * 4 + co];
                    v = fmaf(w00, m[2 * t2], v);
                    v = fmaf(w01, m[2 * t2 + 1], v);
                    v = fmaf(w10, m[8 + 2 * t2], v);
                    v = fmaf(w11, m[8 + 2 * t2 + 1], v);
                    if (v > best) {   // first maximum wins, like max_pool3d
                        best = v;
                        a0 = m[2 * t2]; a1 = m[2 * t2 + 1]; b0 = m[8 + 2 * t2]; b1 = m[8 + 2 * t2 + 1];
                    }
                }
                acc[cl * 5 + 0] = fmaf(gs[k], a0, acc[cl * 5 + 0]);
                acc[cl * 5 + 1] = fmaf(gs[k], a1, acc[cl * 5 + 1]);
                acc[cl * 5 + 2] = fmaf(gs[k], b0, acc[cl * 5 + 2]);
                acc[cl * 5 + 3] = fmaf(gs[k], b1, acc[cl * 5 + 3]);
                acc[cl * 5 + 4] += gs[k];
            }
        }
    }
    // reduce over the 16 pixels of a wave that share a channel group (lanes l, l+4, ...), then over the 4 waves
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 40; ++i) {
        float s = acc[i];
#pragma unroll
        for (int o = 32; o >= 4; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane < 4) {
            const int cl = i / 5, q = i - cl * 5, co = cg * 8 + cl;
            red[wave][(q < 4) ? co * 4 + q : kNF * 4 + co] = s;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kNF * 5; i += 256)
        partial[(long)blockIdx.x * kNF * 5 + i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
}

__global__ void hupr_k_mnet_bwd_final(const float* __restrict__ partial, int nblk, float* __restrict__ dw,
                                      float* __restrict__ dbias) {
    __shared__ double red[4];
    const int i = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) s += partial[(long)b * kNF * 5 + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        s = (red[0] + red[1]) + (red[2] + red[3]);
        if (i < kNF * 4) dw[i] = (float)s;
        else dbias[i - kNF * 4] = (float)s;
    }
}

// ------------------------------------------------------------------------------------------
// linear resampling, align_corners=True (ATen: scale = (in-1)/(out-1) in float, src = scale*dst,
// i0 = (int)src, lambda = src - i0, i1 = min(i0+1, in-1))
// x[b][Di][Hi][Wi][C] (voxel stride in_ld) -> y[b][Do][Ho][Wo][C] (voxel stride out_ld)
// ------------------------------------------------------------------------------------------
struct Lin {
    int i0, i1;
    float w0, w1;
};
__device__ __forceinline__ Lin lin_coord(int o, int in, int out) {
    // src is the ROUNDED product, as in ATen's area_pixel_compute_source_index: fused into the subtraction below (fma) the weights
    // move by up to 2e-6 against the reference's, 1e-5 on the output (HIP's __fmul_rn is a plain, contractable multiply)
#pragma clang fp contract(off)
    Lin l;
    const float scale = (out > 1) ? (float)(in - 1) / (float)(out - 1) : 0.f;
    const float src = scale * (float)o;
    l.i0 = (int)src;
    l.i1 = l.i0 + ((l.i0 < in - 1) ? 1 : 0);
    l.w1 = src - (float)l.i0;
    l.w0 = 1.f - l.w1;
    return l;
}

// contributions of output index o (0..out-1) to input index i: linear weights with align_corners=True.
// For input index i the contributing outputs are those whose i0 == i (weight w0) or i1 == i (weight w1, i1 != i0).
// Because src = scale*o is monotone, they form a contiguous range; we scan a conservative window around i/scale.
__device__ __forceinline__ void lin_range(int i, int in, int out, int& lo, int& hi) {
    if (out == 1 || in == 1) { lo = 0; hi = out - 1; return; }
    const float inv = (float)(out - 1) / (float)(in - 1);
    lo = max(0, (int)floorf((float)(i - 1) * inv) - 1);
    hi = min(out - 1, (int)ceilf((float)(i + 1) * inv) + 1);
}
__device__ __forceinline__ float lin_weight(int o, int i, int in, int out) {
    const Lin l = lin_coord(o, in, out);
    float w = 0.f;
    if (l.i0 == i) w += l.w0;
    if (l.i1 == i) w += l.w1;                            // at the clamped upper edge both taps hit the same voxel
    return w;
}
// the same two helpers with the per-axis ratios hoisted out of the voxel loops (scale = (in-1)/(out-1) exactly as
// lin_coord computes it, inv = (out-1)/(in-1)): the backward kernel evaluates tens of candidate weights per voxel
struct LinAxis {
    int in, out;
    float scale, inv;
};
__device__ __forceinline__ LinAxis lin_axis(int in, int out) {
    LinAxis a;
    a.in = in; a.out = out;
    a.scale = (out > 1) ? (float)(in - 1) / (float)(out - 1) : 0.f;
    a.inv = (in > 1) ? (float)(out - 1) / (float)(in - 1) : 0.f;
    return a;
}
__device__ __forceinline__ void lin_range(int i, const LinAxis& a, int& lo, int& hi) {
    if (a.out == 1 || a.in == 1) { lo = 0; hi = a.out - 1; return; }
    // contributors have scale * o in (i - 1, i + 1), i.e. o in ((i-1) inv, (i+1) inv); 1/32 covers the float rounding of
    // scale, inv and the products for any extent below 2^15 (a zero-weight candidate costs one weight evaluation)
    lo = max(0, (int)ceilf((float)(i - 1) * a.inv - 0.03125f));
    hi = min(a.out - 1, (int)floorf((float)(i + 1) * a.inv + 0.03125f));
}
__device__ __forceinline__ float lin_weight(int o, int i, const LinAxis& a) {
#pragma clang fp contract(off)      // same rounding as lin_coord: the backward is the exact adjoint of the forward
    const float src = a.scale * (float)o;
    const int i0 = (int)src;
    const int i1 = i0 + ((i0 < a.in - 1) ? 1 : 0);
    const float w1 = src - (float)i0;
    float w = 0.f;
    if (i0 == i) w += 1.f - w1;
    if (i1 == i) w += w1;
    return w;
}

// forward: y[o] = sum over the 8 corner voxels
template <typename T, bool PK = false>
__global__ __launch_bounds__(256) void hupr_k_interp_fwd(const T* __restrict__ src, T* __restrict__ dst, int Bn,
                                                         int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C,
                                                         int in_ld, int out_ld) {
    const int c4n = C >> 2;
    const long total = (long)Bn * Do * Ho * Wo * c4n;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c4 = idx % c4n;
        long v = idx / c4n;
        const int ow = v % Wo; v /= Wo;
        const int oh = v % Ho; v /= Ho;
        const int od = v % Do;
        const int b = v / Do;
        const Lin ld = lin_coord(od, Di, Do), lh = lin_coord(oh, Hi, Ho), lw = lin_coord(ow, Wi, Wo);
        const long ovox = (((long)b * Do + od) * Ho + oh) * Wo + ow;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            if (a == 1 && Di == 1) continue;
            const int id = a ? ld.i1 : ld.i0;
            const float wd = a ? ld.w1 : ld.w0;
#pragma unroll
            for (int bq = 0; bq < 2; ++bq) {
                const int ih = bq ? lh.i1 : lh.i0;
                const float wh = bq ? lh.w1 : lh.w0;
#pragma unroll
                for (int cq = 0; cq < 2; ++cq) {
                    const int iw = cq ? lw.i1 : lw.i0;
                    const float wgt = wd * wh * (cq ? lw.w1 : lw.w0);
                    const long ivox = (((long)b * Di + id) * Hi + ih) * Wi + iw;
                    const float4 xv = ld_act4(src + ivox * in_ld + c4 * 4);
#ifdef HUPR_PROBE_WHOLE_SIMD                              // probe builds only: claim all 512 registers -> no other wave shares this wave's SIMD
                    asm volatile("" ::: "v255", "a255");
#endif
                    if constexpr (PK) {                     // probe only (hupr_debug_interp_packed, scripts/interp_race.py builds this file
                        acc.x = fmaf(wgt, xv.x, acc.x); acc.y = fmaf(wgt, xv.y, acc.y);     // WITHOUT -fno-slp-vectorize): the form hipcc's SLP
                        acc.z = fmaf(wgt, xv.z, acc.z); acc.w = fmaf(wgt, xv.w, acc.w);     // vectoriser turns into op_sel'ed v_pk_mul / v_pk_fma
                    } else {
                        // scalar FMAs on purpose.  Under plain -O3 hipcc's SLP vectoriser turns the weight products and the four
                        // accumulations into op_sel'ed v_pk_mul_f32 / v_pk_fma_f32, and with the two encoder branches on two streams that
                        // build returned wrong even-channel sums (2-40 % low, ~1e-4 of the elements) in most launches that shared the chip
                        // with hupr_k_conv_halo_bf16<64, 64> — never alone, not behind poisoned registers, also with full vmcnt(0) waits;
                        // a hand-written v_pk_fma_f32 form does not (scripts/interp_race.py, DESIGN.md section 7 "A two-stream step that
                        // was not reproducible").  The library is built with -fno-slp-vectorize for the same reason.
                        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc.x) : "v"(wgt), "v"(xv.x));
                        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc.y) : "v"(wgt), "v"(xv.y));
                        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc.z) : "v"(wgt), "v"(xv.z));
                        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc.w) : "v"(wgt), "v"(xv.w));
                    }
                }
            }
        }
        st_act4(dst + ovox * out_ld + c4 * 4, acc);
    }
}

// backward, gather form: each thread owns one INPUT voxel x VPT 16-byte channel vectors and sums the output gradients
// that touched it — deterministic, no atomics, no zero-fill pass.  The kernel is VALU-bound on the candidate-weight
// arithmetic (tens of weights per voxel), so a thread covers as many channels of its voxel as the shape allows and the
// candidate window per axis is the exact contributor range widened by a rounding margin only.
// ACC: dx already holds another consumer's gradient of the same tensor (e.g. the temporal merge's): start from it instead of
// zero — the separate gradient-accumulation kernel autograd would run (read 2, write 1 tensor) becomes one extra read here.
template <typename T, int VPT, bool ACC = false>
__global__ __launch_bounds__(256) void hupr_k_interp_bwd(const T* __restrict__ dy, T* __restrict__ dx, int Bn,
                                                         int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C,
                                                         int in_ld, int out_ld) {
    constexpr int V = ActVec<T>::V, CH = V * VPT;
    const int cgn = C / CH;
    const long total = (long)Bn * Di * Hi * Wi * cgn;
    const LinAxis ad = lin_axis(Di, Do), ah = lin_axis(Hi, Ho), aw = lin_axis(Wi, Wo);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c0 = (int)(idx % cgn) * CH;
        long v = idx / cgn;
        const int iw = v % Wi; v /= Wi;
        const int ih = v % Hi; v /= Hi;
        const int id = v % Di;
        const int b = v / Di;
        int dlo, dhi, hlo, hhi, wlo, whi;
        lin_range(id, ad, dlo, dhi);
        lin_range(ih, ah, hlo, hhi);
        lin_range(iw, aw, wlo, whi);
        float acc[CH];
        const long ivox0 = (((long)b * Di + id) * Hi + ih) * Wi + iw;
        if constexpr (ACC) {
#pragma unroll
            for (int u = 0; u < VPT; ++u) ActVec<T>::load(dx + ivox0 * in_ld + c0 + u * V, acc + u * V);
        } else {
#pragma unroll
            for (int k = 0; k < CH; ++k) acc[k] = 0.f;
        }
        for (int od = dlo; od <= dhi; ++od) {
            const float wd = (Di == 1 && Do == 1) ? 1.f : lin_weight(od, id, ad);
            if (wd == 0.f) continue;
            for (int oh = hlo; oh <= hhi; ++oh) {
                const float wh = lin_weight(oh, ih, ah);
                if (wh == 0.f) continue;
                for (int ow = wlo; ow <= whi; ++ow) {
                    const float ww = lin_weight(ow, iw, aw);
                    if (ww == 0.f) continue;
                    const float wgt = wd * wh * ww;
                    const long ovox = (((long)b * Do + od) * Ho + oh) * Wo + ow;
                    const T* src = dy + ovox * out_ld + c0;
#pragma unroll
                    for (int u = 0; u < VPT; ++u) {
                        float g[V];
                        ActVec<T>::load(src + u * V, g);
#pragma unroll
                        for (int k = 0; k < V; ++k) acc[u * V + k] = fmaf(wgt, g[k], acc[u * V + k]);
                    }
                }
            }
        }
        const long ivox = (((long)b * Di + id) * Hi + ih) * Wi + iw;
#pragma unroll
        for (int u = 0; u < VPT; ++u) ActVec<T>::store(dx + ivox * in_ld + c0 + u * V, acc + u * V);
    }
}

}  // namespace hupr

using namespace hupr;

// (a3) MNet forward.  x: (n_bg = B*G, F=8, 2, pixels = R*A, E=8) fp32;  w: (32,2,2,1,1); bias: (32)
// out: (n_bg, pixels, 32) channels-last, fp32 or (bf16act) bf16
template <typename T>
static int mnet_fwd(const char* who, const float* x, const float* w, const float* bias, T* out, float* means, long n_bg,
                    int pixels, hupr_stream_t stream) {
    HUPR_REQUIRE(x && w && bias && out && n_bg > 0 && pixels > 0, "%s: bad argument", who);
    const long total = n_bg * pixels;
    const int grid = (int)min((long)8192, (total + 255) / 256);
    HUPR_LAUNCH(hupr_k_mnet_fwd<T>, dim3(grid), dim3(256), 0, as_stream(stream), x, w, bias, out, means, n_bg, pixels);
    HUPR_LAUNCH_OK("hupr_k_mnet_fwd");
    return HUPR_OK;
}
extern "C" int hupr_mnet_fwd_f32(const float* x, const float* w, const float* bias, float* out, float* means_or_null,
                                 long n_bg, int pixels, hupr_stream_t stream) {
    return mnet_fwd("hupr_mnet_fwd_f32", x, w, bias, out, means_or_null, n_bg, pixels, stream);
}
extern "C" int hupr_mnet_fwd_bf16act(const float* x, const float* w, const float* bias, void* out, float* means_or_null,
                                     long n_bg, int pixels, hupr_stream_t stream) {
    return mnet_fwd("hupr_mnet_fwd_bf16act", x, w, bias, static_cast<__bf16*>(out), means_or_null, n_bg, pixels, stream);
}

template <typename T>
static int mnet_fwd_means(const char* who, const float* mp, const float* w, const float* bias, T* out, float* means, long n_bg,
                          int pixels, hupr_stream_t stream) {
    HUPR_REQUIRE(mp && w && bias && out && n_bg > 0 && pixels > 0, "%s: bad argument", who);
    const long total = n_bg * pixels;
    const int grid = (int)min((long)8192, (total + 255) / 256);
    HUPR_LAUNCH(hupr_k_mnet_fwd_means<T>, dim3(grid), dim3(256), 0, as_stream(stream), mp, w, bias, out, means, n_bg, pixels);
    HUPR_LAUNCH_OK("hupr_k_mnet_fwd_means");
    return HUPR_OK;
}
extern "C" int hupr_mnet_fwd_means_f32(const float* mean_planes, const float* w, const float* bias, float* out,
                                       float* means_or_null, long n_bg, int pixels, hupr_stream_t stream) {
    return mnet_fwd_means("hupr_mnet_fwd_means_f32", mean_planes, w, bias, out, means_or_null, n_bg, pixels, stream);
}
extern "C" int hupr_mnet_fwd_means_bf16act(const float* mean_planes, const float* w, const float* bias, void* out,
                                           float* means_or_null, long n_bg, int pixels, hupr_stream_t stream) {
    return mnet_fwd_means("hupr_mnet_fwd_means_bf16act", mean_planes, w, bias, static_cast<__bf16*>(out), means_or_null, n_bg,
                          pixels, stream);
}

extern "C" size_t hupr_mnet_bwd_ws_bytes(void) { return (size_t)1024 * kNF * 5 * sizeof(float); }

template <typename T>
static int mnet_bwd(const char* who, const float* x, const float* means, const float* w, const float* bias, const T* dy,
                    float* dw, float* dbias, long n_bg, int pixels, void* ws, size_t ws_bytes, hupr_stream_t stream) {
    HUPR_REQUIRE((x || means) && w && bias && dy && dw && dbias && ws && n_bg > 0 && pixels > 0, "%s: bad argument", who);
    if (ws_bytes < hupr_mnet_bwd_ws_bytes()) return fail(HUPR_ERR_WORKSPACE, "%s: workspace too small", who);
    const long total = n_bg * pixels;
    const int grid = (int)min((long)1024, (total + 63) / 64);      // 64 pixels per workgroup pass; partial[grid][160]
    hipStream_t s = as_stream(stream);
    HUPR_LAUNCH(hupr_k_mnet_bwd<T>, dim3(grid), dim3(256), 0, s, x, w, bias, dy, means, n_bg, pixels,
                       reinterpret_cast<float*>(ws));
    HUPR_LAUNCH_OK("hupr_k_mnet_bwd");
    HUPR_LAUNCH(hupr_k_mnet_bwd_final, dim3(kNF * 5), dim3(256), 0, s, reinterpret_cast<const float*>(ws), grid, dw, dbias);
    HUPR_LAUNCH_OK("hupr_k_mnet_bwd_final");
    return HUPR_OK;
}
extern "C" int hupr_mnet_bwd_f32(const float* x_or_null, const float* means_or_null, const float* w, const float* bias,
                                 const float* dy, float* dw, float* dbias, long n_bg, int pixels, void* ws, size_t ws_bytes,
                                 hupr_stream_t stream) {
    return mnet_bwd("hupr_mnet_bwd_f32", x_or_null, means_or_null, w, bias, dy, dw, dbias, n_bg, pixels, ws, ws_bytes, stream);
}
extern "C" int hupr_mnet_bwd_bf16act(const float* x_or_null, const float* means_or_null, const float* w, const float* bias,
                                     const void* dy, float* dw, float* dbias, long n_bg, int pixels, void* ws, size_t ws_bytes,
                                     hupr_stream_t stream) {
    return mnet_bwd("hupr_mnet_bwd_bf16act", x_or_null, means_or_null, w, bias, static_cast<const __bf16*>(dy), dw, dbias, n_bg,
                    pixels, ws, ws_bytes, stream);
}

static int interp_check(const char* who, int Bn, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C, int in_ld,
                        int out_ld) {
    HUPR_REQUIRE(Bn > 0 && Di > 0 && Hi > 0 && Wi > 0 && Do > 0 && Ho > 0 && Wo > 0, "%s: bad extents", who);
    HUPR_REQUIRE(C > 0 && C % 4 == 0 && in_ld % 4 == 0 && out_ld % 4 == 0 && in_ld >= C && out_ld >= C,
                 "%s: channels/strides must be multiples of 4 (C=%d in_ld=%d out_ld=%d)", who, C, in_ld, out_ld);
    return HUPR_OK;
}

template <typename T>
static int interp_fwd(const char* who, const T* x, T* y, int Bn, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C,
                      int in_ld, int out_ld, hupr_stream_t stream) {
    HUPR_REQUIRE(x && y, "%s: null pointer", who);
    int rc = interp_check(who, Bn, Di, Hi, Wi, Do, Ho, Wo, C, in_ld, out_ld);
    if (rc) return rc;
    const long total = (long)Bn * Do * Ho * Wo * (C / 4);
    HUPR_LAUNCH((hupr_k_interp_fwd<T, false>), dim3((int)min((long)8192, (total + 255) / 256)), dim3(256), 0,
                       as_stream(stream), x, y, Bn, Di, Hi, Wi, Do, Ho, Wo, C, in_ld, out_ld);
    HUPR_LAUNCH_OK("hupr_k_interp_fwd");
    return HUPR_OK;
}
extern "C" int hupr_interp_linear_fwd_f32(const float* x, float* y, int Bn, int Di, int Hi, int Wi, int Do, int Ho,
                                          int Wo, int C, int in_ld, int out_ld, hupr_stream_t stream) {
    return interp_fwd("hupr_interp_linear_fwd_f32", x, y, Bn, Di, Hi, Wi, Do, Ho, Wo, C, in_ld, out_ld, stream);
}
extern "C" int hupr_interp_linear_fwd_bf16act(const void* x, void* y, int Bn, int Di, int Hi, int Wi, int Do, int Ho,
                                              int Wo, int C, int in_ld, int out_ld, hupr_stream_t stream) {
    return interp_fwd("hupr_interp_linear_fwd_bf16act", static_cast<const __bf16*>(x), static_cast<__bf16*>(y), Bn, Di,
                      Hi, Wi, Do, Ho, Wo, C, in_ld, out_ld, stream);
}

// dx = adjoint of the forward map applied to dy (gather form: deterministic, every dx voxel written once)
template <typename T, bool ACC = false>
static int interp_bwd(const char* who, const T* dy, T* dx, int Bn, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C,
                      int in_ld, int out_ld, hupr_stream_t stream) {
    HUPR_REQUIRE(dy && dx, "%s: null pointer", who);
    int rc = interp_check(who, Bn, Di, Hi, Wi, Do, Ho, Wo, C, in_ld, out_ld);
    if (rc) return rc;
    constexpr int V = ActVec<T>::V;
    HUPR_REQUIRE(C % V == 0 && in_ld % V == 0 && out_ld % V == 0, "%s: channels / strides must be multiples of %d", who, V);
    const long voxels = (long)Bn * Di * Hi * Wi;
    if (C % (4 * V) == 0 && voxels * (C / (4 * V)) >= 256 * 256) {       // 4 vectors per thread while the grid stays full
        const long total = voxels * (C / (4 * V));
        HUPR_LAUNCH((hupr_k_interp_bwd<T, 4, ACC>), dim3((int)min((long)16384, (total + 255) / 256)), dim3(256), 0,
                           as_stream(stream), dy, dx, Bn, Di, Hi, Wi, Do, Ho, Wo, C, in_ld, out_ld);
    } else {
        const long total = voxels * (C / V);
        HUPR_LAUNCH((hupr_k_interp_bwd<T, 1, ACC>), dim3((int)min((long)16384, (total + 255) / 256)), dim3(256), 0,
                           as_stream(stream), dy, dx, Bn, Di, Hi, Wi, Do, Ho, Wo, C, in_ld, out_ld);
    }
    HUPR_LAUNCH_OK("hupr_k_interp_bwd");
    return HUPR_OK;
}
extern "C" int hupr_interp_linear_bwd_f32(const float* dy, float* dx, int Bn, int Di, int Hi, int Wi, int Do, int Ho,
                                          int Wo, int C, int in_ld, int out_ld, hupr_stream_t stream) {
    return interp_bwd("hupr_interp_linear_bwd_f32", dy, dx, Bn, Di, Hi, Wi, Do, Ho, Wo, C, in_ld, out_ld, stream);
}
extern "C" int hupr_interp_linear_bwd_bf16act(const void* dy, void* dx, int Bn, int Di, int Hi, int Wi, int Do, int Ho,
                                              int Wo, int C, int in_ld, int out_ld, hupr_stream_t stream) {
    return interp_bwd("hupr_interp_linear_bwd_bf16act", static_cast<const __bf16*>(dy), static_cast<__bf16*>(dx), Bn, Di,
                      Hi, Wi, Do, Ho, Wo, C, in_ld, out_ld, stream);
}

// dx += the input gradient (dx holds another consumer's gradient of the same tensor on entry)
extern "C" int hupr_interp_linear_bwd_acc_f32(const float* dy, float* dx, int Bn, int Di, int Hi, int Wi, int Do, int Ho,
                                              int Wo, int C, int in_ld, int out_ld, hupr_stream_t stream) {
    return interp_bwd<float, true>("hupr_interp_linear_bwd_acc_f32", dy, dx, Bn, Di, Hi, Wi, Do, Ho, Wo, C, in_ld, out_ld, stream);
}
extern "C" int hupr_interp_linear_bwd_acc_bf16act(const void* dy, void* dx, int Bn, int Di, int Hi, int Wi, int Do, int Ho,
                                                  int Wo, int C, int in_ld, int out_ld, hupr_stream_t stream) {
    return interp_bwd<__bf16, true>("hupr_interp_linear_bwd_acc_bf16act", static_cast<const __bf16*>(dy), static_cast<__bf16*>(dx), Bn,
                                    Di, Hi, Wi, Do, Ho, Wo, C, in_ld, out_ld, stream);
}
