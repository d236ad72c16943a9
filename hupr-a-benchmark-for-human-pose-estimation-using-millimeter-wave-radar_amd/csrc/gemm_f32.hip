// fp32 MFMA GEMM / implicit-GEMM convolution engine for gfx950.
//
// One templated kernel serves every GEMM-shaped op of HuPRNet in exact
// fp32 (v_mfma_f32_32x32x2_f32: bit-equal to an fmaf chain, 157 TF peak):
//
//   A operand modes                         B operand modes
//     A_ROWK   A[m][k], k contiguous          B_NK     B[n][k], k contiguous  (packed weights, "NT")
//     A_CONV   im2col gather of a              B_KN     B[k][n], n contiguous  ("NN")
//              channels-last activation        B_CONVK  im2col gather, K-major: k = voxel,
//     A_KM     A[k][m], m contiguous ("T")              n = (tap, ci)           (weight gradient)
//
//   conv fwd / dgrad / 1x1 / temporal merge : A_CONV x B_NK     (+bias, +residual epilogue)
//   attention S = Q K^T, dP = dO V^T        : A_ROWK x B_NK     (batched)
//   attention O = P V, dQ = dS K, GCN W x   : A_ROWK x B_KN     (batched)
//   attention dV = P^T dO, dK = dS^T Q      : A_KM   x B_KN     (batched)
//   conv weight gradient                    : A_KM   x B_CONVK  (split over the voxel axis)
//
// Tiling: 256 threads = 4 waves; block tile BM x BN x 32; each wave owns (BM/WM) x (BN/WN) as
// 32x32 MFMA tiles; operands are staged global -> registers -> LDS with the next tile's global
// loads in flight during the MFMA phase.  LDS images are chosen per mode so that both the
// staging stores and the one-float-per-lane MFMA operand reads are bank-conflict free
// (row stride 33 for k-contiguous tiles, dense rows for k-major tiles).
#include "gemm_common.h"

namespace hupr {

typedef float f32x4n __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldv4(const float* p) {   // native vector load: stays one global_load_dwordx4
    const f32x4n t = *reinterpret_cast<const f32x4n*>(p);
    return make_float4(t.x, t.y, t.z, t.w);
}

constexpr int BK = 32;

template <int BM, int BN, int WM, int WN, int AM, int BMD>
__global__ __launch_bounds__(256) void hupr_k_gemm_f32(GemmArgs p) {
    constexpr int WTM = BM / WM, WTN = BN / WN;      // wave tile
    constexpr int TM = WTM / 32, TN = WTN / 32;      // 32x32 MFMA tiles per wave
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "bad tiling");
    constexpr int A_LD = (AM == A_KM) ? BM : BK + 1;      // LDS row stride (floats)
    constexpr int B_LD = (BMD == B_NK) ? BK + 1 : BN;
    constexpr int A_ROWS = (AM == A_KM) ? BK : BM;
    constexpr int B_ROWS = (BMD == B_NK) ? BN : BK;
    constexpr int A_F4 = BM * BK / 4 / 256;          // float4 per thread per tile
    constexpr int B_F4 = BN * BK / 4 / 256;
    static_assert(A_F4 >= 1 && B_F4 >= 1, "tile too small for 256 threads");

    __shared__ float As[A_ROWS * A_LD];
    __shared__ float Bs[B_ROWS * B_LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // ---- block coordinates -------------------------------------------------------------
    const int n_tiles = (p.N + BN - 1) / BN;
    // XCD-aware order: workgroup b runs on XCD b % 8 (observed), so give each XCD a contiguous run of
    // tiles — neighbouring output tiles share im2col halos / operand panels through that XCD's L2.
    int tile = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = tile & 7, idx = tile >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (tile / n_tiles) * BM, n0 = (tile % n_tiles) * BN;
    const int z = blockIdx.z, z0 = z / p.zdiv, z1 = z % p.zdiv;
    const float* __restrict__ Ag = p.A + z0 * p.a_bs0 + z1 * p.a_bs1;
    const float* __restrict__ Bg = p.B + z0 * p.b_bs0 + z1 * p.b_bs1;
    float* __restrict__ Cg = p.C + z0 * p.c_bs0 + z1 * p.c_bs1;

    // K range of this block (split over blockIdx.y)
    const int k_tiles_total = (p.K + BK - 1) / BK;
    const int per = (k_tiles_total + p.ksplit - 1) / p.ksplit;
    const int kt_begin = blockIdx.y * per;
    const int kt_end = min(k_tiles_total, kt_begin + per);
    if (p.ksplit > 1) Cg += (long)blockIdx.y * p.split_stride;

    const ConvGeom& g = p.g;

    // ---- per-thread staging descriptors ---------------------------------------------------
    // A_ROWK / A_CONV: float4 index f = tid + 256*i -> row = f / 8, k4 = f % 8
    // A_KM           : f -> krow = f / (BM/4), c4 = f % (BM/4)
    float4 ra[A_F4], rb[B_F4];
    long a_off[A_F4];        // A_ROWK: row offset; A_CONV: base offset of output voxel's (0,0,0) tap
    int a_vox[A_F4];         // A_CONV: packed (od, oh, ow) ; -1 if row out of range
    if (AM == A_ROWK) {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int row = m0 + (tid + 256 * i) / 8;
            a_off[i] = (row < p.M) ? (long)row * p.lda : -1;
        }
    } else if (AM == A_CONV) {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int row = m0 + (tid + 256 * i) / 8;
            if (row < p.M) {
                int ow = row % g.Wo, t = row / g.Wo;
                int oh = t % g.Ho; t /= g.Ho;
                int od = t % g.Do, b = t / g.Do;
                a_vox[i] = (od << 20) | (oh << 10) | ow;
                a_off[i] = (long)b * g.Di * g.Hi * g.Wi;
            } else {
                a_vox[i] = -1;
                a_off[i] = 0;
            }
        }
    }

    auto load_a = [&](int kt) {
        const int k0 = kt * BK;
        if (AM == A_ROWK) {
#pragma unroll
            for (int i = 0; i < A_F4; ++i) {
                const int k = k0 + ((tid + 256 * i) & 7) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a_off[i] >= 0) {
                    const float* src = Ag + a_off[i] + k;
                    if (k + 3 < p.K && ((p.lda & 3) == 0)) {
                        v = ldv4(src);
                    } else {
                        if (k < p.K) v.x = src[0];
                        if (k + 1 < p.K) v.y = src[1];
                        if (k + 2 < p.K) v.z = src[2];
                        if (k + 3 < p.K) v.w = src[3];
                    }
                }
                ra[i] = v;
            }
        } else if (AM == A_CONV) {
            // whole k-tile lies inside one tap because Ci % 32 == 0
            const int tap = k0 / g.Ci, ci0 = k0 - tap * g.Ci;
            const int tw_ = tap % g.kw, tt = tap / g.kw;
            const int th_ = tt % g.kh, td_ = tt / g.kh;
#pragma unroll
            for (int i = 0; i < A_F4; ++i) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a_vox[i] >= 0) {
                    const int id = (a_vox[i] >> 20) + td_ - g.pd;
                    const int ih = ((a_vox[i] >> 10) & 1023) + th_ - g.ph;
                    const int iw = (a_vox[i] & 1023) + tw_ - g.pw;
                    if ((unsigned)id < (unsigned)g.Di && (unsigned)ih < (unsigned)g.Hi &&
                        (unsigned)iw < (unsigned)g.Wi) {
                        const long vox = a_off[i] + ((long)id * g.Hi + ih) * g.Wi + iw;
                        v = *reinterpret_cast<const float4*>(Ag + vox * g.in_ld + ci0 +
                                                             ((tid + 256 * i) & 7) * 4);
                    }
                }
                ra[i] = v;
            }
        } else {   // A_KM: A[k][m]
#pragma unroll
            for (int i = 0; i < A_F4; ++i) {
                const int f = tid + 256 * i;
                const int k = k0 + f / (BM / 4), m = m0 + (f % (BM / 4)) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < p.K) {
                    const float* src = Ag + (long)k * p.lda + m;
                    if (m + 3 < p.M && ((p.lda & 3) == 0)) {
                        v = ldv4(src);
                    } else {
                        if (m < p.M) v.x = src[0];
                        if (m + 1 < p.M) v.y = src[1];
                        if (m + 2 < p.M) v.z = src[2];
                        if (m + 3 < p.M) v.w = src[3];
                    }
                }
                ra[i] = v;
            }
        }
    };

    auto load_b = [&](int kt) {
        const int k0 = kt * BK;
        if (BMD == B_NK) {
#pragma unroll
            for (int i = 0; i < B_F4; ++i) {
                const int f = tid + 256 * i;
                const int n = n0 + f / 8, k = k0 + (f & 7) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n < p.N) {
                    const float* src = Bg + (long)n * p.ldb + k;
                    if (k + 3 < p.K && ((p.ldb & 3) == 0)) {
                        v = ldv4(src);
                    } else {
                        if (k < p.K) v.x = src[0];
                        if (k + 1 < p.K) v.y = src[1];
                        if (k + 2 < p.K) v.z = src[2];
                        if (k + 3 < p.K) v.w = src[3];
                    }
                }
                rb[i] = v;
            }
        } else if (BMD == B_KN) {
#pragma unroll
            for (int i = 0; i < B_F4; ++i) {
                const int f = tid + 256 * i;
                const int k = k0 + f / (BN / 4), n = n0 + (f % (BN / 4)) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < p.K) {
                    const float* src = Bg + (long)k * p.ldb + n;
                    if (n + 3 < p.N && ((p.ldb & 3) == 0)) {
                        v = ldv4(src);
                    } else {
                        if (n < p.N) v.x = src[0];
                        if (n + 1 < p.N) v.y = src[1];
                        if (n + 2 < p.N) v.z = src[2];
                        if (n + 3 < p.N) v.w = src[3];
                    }
                }
                rb[i] = v;
            }
        } else {   // B_CONVK: k = output voxel index, n = tap*Ci + ci (a tile may span several taps)
#pragma unroll
            for (int i = 0; i < B_F4; ++i) {
                const int f = tid + 256 * i;
                const int k = k0 + f / (BN / 4);
                const int n = n0 + (f % (BN / 4)) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < p.K && n < p.N) {
                    const int tap = n / g.Ci, ci = n - tap * g.Ci;
                    const int tw_ = tap % g.kw, tt = tap / g.kw;
                    const int th_ = tt % g.kh, td_ = tt / g.kh;
                    int ow = k % g.Wo, t = k / g.Wo;
                    int oh = t % g.Ho; t /= g.Ho;
                    int od = t % g.Do, b = t / g.Do;
                    const int id = od + td_ - g.pd, ih = oh + th_ - g.ph, iw = ow + tw_ - g.pw;
                    if ((unsigned)id < (unsigned)g.Di && (unsigned)ih < (unsigned)g.Hi &&
                        (unsigned)iw < (unsigned)g.Wi) {
                        const long vox = (((long)b * g.Di + id) * g.Hi + ih) * g.Wi + iw;
                        v = *reinterpret_cast<const float4*>(Bg + vox * g.in_ld + ci);
                    }
                }
                rb[i] = v;
            }
        }
    };

    auto store_tiles = [&]() {
        if (AM == A_KM) {
#pragma unroll
            for (int i = 0; i < A_F4; ++i) {
                const int f = tid + 256 * i;
                *reinterpret_cast<float4*>(&As[(f / (BM / 4)) * A_LD + (f % (BM / 4)) * 4]) = ra[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_F4; ++i) {
                const int f = tid + 256 * i;
                float* d = &As[(f / 8) * A_LD + (f & 7) * 4];
                d[0] = ra[i].x; d[1] = ra[i].y; d[2] = ra[i].z; d[3] = ra[i].w;
            }
        }
        if (BMD == B_NK) {
#pragma unroll
            for (int i = 0; i < B_F4; ++i) {
                const int f = tid + 256 * i;
                float* d = &Bs[(f / 8) * B_LD + (f & 7) * 4];
                d[0] = rb[i].x; d[1] = rb[i].y; d[2] = rb[i].z; d[3] = rb[i].w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < B_F4; ++i) {
                const int f = tid + 256 * i;
                *reinterpret_cast<float4*>(&Bs[(f / (BN / 4)) * B_LD + (f % (BN / 4)) * 4]) = rb[i];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lr = lane & 31, lh = lane >> 5;
    if (kt_begin < kt_end) {
        load_a(kt_begin);
        load_b(kt_begin);
    }
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        store_tiles();
        __syncthreads();
        if (kt + 1 < kt_end) {
            load_a(kt + 1);
            load_b(kt + 1);
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * WTM + i * 32 + lr;
                a[i] = (AM == A_KM) ? As[(kk + lh) * A_LD + row] : As[row * A_LD + kk + lh];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = wn * WTN + j * 32 + lr;
                b[j] = (BMD == B_NK) ? Bs[col * B_LD + kk + lh] : Bs[(kk + lh) * B_LD + col];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: D[i][j]: j = lane&31, i = (r&3) + 8*(r>>2) + 4*(lane>>5) ------------------
    const float* __restrict__ resg = p.res ? p.res + z0 * p.res_bs0 + z1 * p.res_bs1 : nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * WTN + j * 32 + lr;
            if (col >= p.N) continue;
            const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < p.M) {
                    float v = acc[i][j][r] + bv;
                    if (resg) v += resg[(long)row * p.res_ld + col];
                    float* dst = Cg + (long)row * p.ldc + col;
                    if (p.accumulate) v += *dst;
                    *dst = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// split-K reduction (deterministic): out[i] (+)= sum_s part[s][i]; optional layout change for
// conv weight gradients: partial is [Cout][taps][Ci] (GEMM order), destination is the
// parameter layout [Cout][Ci][taps] (PyTorch (Cout,Cin,kd,kh,kw)).
// ------------------------------------------------------------------------------------------
__global__ void hupr_k_splitk_reduce(const float* __restrict__ part, float* __restrict__ out, long n,
                                     int splits, long split_stride, int taps, int ci) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += part[k * split_stride + i];
    long dst = i;
    if (taps > 1) {
        const long per = (long)taps * ci;
        const long co = i / per, rem = i - co * per;
        const int tap = rem / ci, c = rem - (long)tap * ci;
        dst = co * per + (long)c * taps + tap;
    }
    out[dst] = s;
}

// weight packing: w (Cout, Cin, taps) ->
//   mode 0 (forward) : wp[co][tap][ci]
//   mode 1 (dgrad)   : wp[ci][taps-1-tap][co]   (flipped taps, in/out swapped)
__global__ void hupr_k_pack_weights(const float* __restrict__ w, float* __restrict__ wp, int co_n,
                                    int ci_n, int taps, int mode) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long n = (long)co_n * ci_n * taps;
    if (i >= n) return;
    // i indexes the destination (coalesced writes)
    if (mode == 0) {
        const int ci = i % ci_n;
        long t = i / ci_n;
        const int tap = t % taps;
        const int co = t / taps;
        wp[i] = w[((long)co * ci_n + ci) * taps + tap];
    } else {
        const int co = i % co_n;
        long t = i / co_n;
        const int tapf = t % taps;
        const int ci = t / taps;
        wp[i] = w[((long)co * ci_n + ci) * taps + (taps - 1 - tapf)];
    }
}

// ------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int AM, int BMD>
static void launch(const GemmArgs& a, int batch, hipStream_t s) {
    const int mt = (a.M + BM - 1) / BM, nt = (a.N + BN - 1) / BN;
    dim3 grid(mt * nt, a.ksplit, batch);
    HUPR_LAUNCH((hupr_k_gemm_f32<BM, BN, WM, WN, AM, BMD>), grid, dim3(256), 0, s, a);
}

template <int AM, int BMD>
static void dispatch_tiles(const GemmArgs& a, int batch, hipStream_t s) {
    // largest tile that still gives >= ~2 workgroups per CU (256 CUs); N tile from N
    auto blocks = [&](int bm, int bn) { return (long)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn) * batch; };
    if (a.N > 64) {
        if (a.M > 64 && blocks(128, 128) >= 512) launch<128, 128, 2, 2, AM, BMD>(a, batch, s);
        else launch<64, 128, 1, 4, AM, BMD>(a, batch, s);
    } else if (a.N > 32) {
        if (a.M > 64 && blocks(128, 64) >= 512) launch<128, 64, 2, 2, AM, BMD>(a, batch, s);
        else launch<64, 64, 2, 2, AM, BMD>(a, batch, s);
    } else {
        launch<128, 32, 4, 1, AM, BMD>(a, batch, s);
    }
}

// Four outputs per thread (16-byte loads), the split axis cut into S slices per workgroup (256 / S columns of four outputs each) and
// EIGHT loads of a thread in flight: the kernel is a chain of dependent memory round trips otherwise (round 5: a layer with 128
// partial tensors and 4 slices walked 32 loads four at a time = 8 round trips, 8 us for 64 workgroups' worth of output; with 16 slices
// — four times the workgroups, 8 loads per thread, one round trip — the small layers take what a launch takes).  Fixed summation
// order (per thread in k order over eight accumulators, their tree, then the slices in order) -> deterministic.
template <int S>
__global__ __launch_bounds__(256) void hupr_k_splitk_reduce4(const float* __restrict__ part, float* __restrict__ out,
                                                             long n4, int splits, long split_stride, int taps, int ci,
                                                             float* __restrict__ out2, long n_first) {
    // out2: elements [n_first, 4 n4) of the summed tensor go to out2 (a second weight gradient of the same shape: rows of the second
    // half of the output channels), laid out like the first
    constexpr int COLS = 256 / S, U = 8;
    __shared__ float4 red[S][COLS];
    const int col = threadIdx.x % COLS, sl = threadIdx.x / COLS;
    const long i4 = (long)blockIdx.x * COLS + col;
    float4 a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i4 < n4) {
        const float* p = part + i4 * 4;
        for (int k = sl; k < splits; k += S * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                v[u] = (k + u * S < splits) ? *reinterpret_cast<const float4*>(p + (long)(k + u * S) * split_stride)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < U; ++u) { a[u].x += v[u].x; a[u].y += v[u].y; a[u].z += v[u].z; a[u].w += v[u].w; }
        }
    }
#pragma unroll
    for (int w = 1; w < U; w *= 2)
#pragma unroll
        for (int u = 0; u + w < U; u += 2 * w) { a[u].x += a[u + w].x; a[u].y += a[u + w].y; a[u].z += a[u + w].z; a[u].w += a[u + w].w; }
    red[sl][col] = a[0];
    __syncthreads();
    if (sl != 0 || i4 >= n4) return;
    float4 r = red[0][col];
#pragma unroll
    for (int q = 1; q < S; ++q) { const float4 t = red[q][col]; r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
    const float s4[4] = {r.x, r.y, r.z, r.w};
    long i = i4 * 4;
    if (out2 != nullptr && i >= n_first) { i -= n_first; out = out2; }
    if (taps > 1) {                                  // [co][tap][ci] -> parameter layout [co][ci][tap]
        const long per = (long)taps * ci;
        const long co = i / per, rem = i - co * per;
        const int tap = rem / ci, c = rem - (long)tap * ci;
#pragma unroll
        for (int j = 0; j < 4; ++j) out[co * per + (long)(c + j) * taps + tap] = s4[j];
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) out[i + j] = s4[j];      // out may be a 4-byte aligned bucket view
    }
}

// The same sums with the parameter-layout write made contiguous (round 6).  hupr_k_splitk_reduce4 stores the four
// sums of a thread 4 bytes each, taps * 4 bytes apart ([co][tap][ci] -> [co][ci][tap]): a wave's store instruction touches 64 cache
// lines, and the level-3 / decoder gradients (1.8-4.7 M elements, 5-13 partial tensors) ran at 2 TB/s of reads.  Here a workgroup
// owns ONE output row segment — output channel co, CB4 * 4 consecutive input channels, all taps: taps runs of CB4 float4 in the
// partial layout (64- or 128-byte runs, coalesced), ONE contiguous run of CB4 * 4 * taps floats in the parameter layout — sums every
// (element, slice) pair with exactly the arithmetic of hupr_k_splitk_reduce4<S> (eight accumulators in k order, their tree, the slices in
// order: bit-identical results), transposes the sums through LDS and writes the run with consecutive 4-byte stores of consecutive
// lanes.  S / CB4 = 4 / 8: the 4-slice class of launch_splitk_reduce (its slice rule is unchanged).
template <int S, int CB4>
__global__ __launch_bounds__(256) void hupr_k_splitk_reduce_t(const float* __restrict__ part, float* __restrict__ out, int splits,
                                                              long split_stride, int taps, int ci, int n_ci_blocks,
                                                              float* __restrict__ out2, long n_first) {
    constexpr int U = 8, CB = CB4 * 4, MAXT = 27;
    __shared__ float4 red[S][MAXT * CB4];
    __shared__ float tr[CB * MAXT + 1];
    const int row = blockIdx.x / n_ci_blocks, cblk = blockIdx.x - row * n_ci_blocks;       // row = output channel (of both gradients)
    const int items = taps * CB4;                              // float4 sums of this workgroup: (tap, column c4)
    const long per = (long)taps * ci;
    const float* prow = part + row * per + cblk * CB;
    for (int p = threadIdx.x; p < items * S; p += 256) {
        const int sl = p / items, it = p - sl * items;
        const int tap = it / CB4, c4 = it - tap * CB4;
        const float* q = prow + (long)tap * ci + c4 * 4;
        float4 a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = sl; k < splits; k += S * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                v[u] = (k + u * S < splits) ? *reinterpret_cast<const float4*>(q + (long)(k + u * S) * split_stride)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < U; ++u) { a[u].x += v[u].x; a[u].y += v[u].y; a[u].z += v[u].z; a[u].w += v[u].w; }
        }
#pragma unroll
        for (int w = 1; w < U; w *= 2)
#pragma unroll
            for (int u = 0; u + w < U; u += 2 * w) { a[u].x += a[u + w].x; a[u].y += a[u + w].y; a[u].z += a[u + w].z; a[u].w += a[u + w].w; }
        red[sl][it] = a[0];
    }
    __syncthreads();
    for (int it = threadIdx.x; it < items; it += 256) {
        float4 r = red[0][it];
#pragma unroll
        for (int q = 1; q < S; ++q) { const float4 t = red[q][it]; r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
        const int tap = it / CB4, c = (it - tap * CB4) * 4;
        tr[(c + 0) * taps + tap] = r.x;                        // [channel of the block][tap]: the parameter layout of the run
        tr[(c + 1) * taps + tap] = r.y;
        tr[(c + 2) * taps + tap] = r.z;
        tr[(c + 3) * taps + tap] = r.w;
    }
    __syncthreads();
    long i = row * per;                                        // first element of this row in the summed tensor
    float* o = out;
    if (out2 != nullptr && i >= n_first) { i -= n_first; o = out2; }
    o += i + (long)cblk * CB * taps;
    for (int e = threadIdx.x; e < CB * taps; e += 256) o[e] = tr[e];      // out may be a 4-byte aligned bucket view
}

static int g_splitk_slices = 0;      // test aid (hupr_debug_splitk_slices): 0 auto, 4 / 16 forced; + 256: the scattered-store kernel of rounds 1-5
extern "C" void hupr_debug_splitk_slices(int s) { g_splitk_slices = s; }

void launch_splitk_reduce(const float* part, float* out, long n, int splits, long split_stride, int taps, int ci,
                          hipStream_t s, float* out2, long n_first) {
    if (out2 != nullptr && !(n % 4 == 0 && ci % 4 == 0 && split_stride % 4 == 0 && n_first % 4 == 0 && (reinterpret_cast<uintptr_t>(part) & 15) == 0)) {
        launch_splitk_reduce(part, out, n_first, splits, split_stride, taps, ci, s, nullptr, 0);         // two plain reductions
        launch_splitk_reduce(part + n_first, out2, n - n_first, splits, split_stride, taps, ci, s, nullptr, 0);
        return;
    }
    if (n % 4 == 0 && ci % 4 == 0 && split_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(part) & 15) == 0) {
        const long n4 = n / 4;
        // many partial tensors over a small output: 16 slices (a quarter of the columns per workgroup, four times the workgroups)
        const long n4_one = out2 ? n_first / 4 : n4;          // (two gradients in one tensor: the choice each of them gets alone)
        const int forced = g_splitk_slices & 255;
        const bool s16 = forced ? forced == 16 : (splits >= 32 && n4_one <= (1L << 17));
        // convolution weight gradients ([co][tap][ci] partials) of the 4-slice class — the large gradients with few partial tensors
        // (levels 2-3, the decoder) —: the contiguous-store kernel where its row segments tile the tensor.  (The 16-slice class —
        // many partials over a small output — stays on hupr_k_splitk_reduce4<16>: a workgroup of the row kernel would walk 7
        // sequential rounds of (element, slice) pairs there; measured 274 vs 208 us per step, profiles/r06_splitk_ab.txt.)
        constexpr int cb = 32;
        if (!s16 && taps > 1 && taps <= 27 && ci % cb == 0 && !(g_splitk_slices & 256) && (n / ((long)taps * ci)) * ((long)taps * ci) == n &&
            (out2 == nullptr || n_first % ((long)taps * ci) == 0)) {
            const int rows = (int)(n / ((long)taps * ci)), nb = ci / cb;
            HUPR_LAUNCH((hupr_k_splitk_reduce_t<4, 8>), dim3(rows * nb), dim3(256), 0, s, part, out, splits, split_stride, taps, ci, nb, out2, n_first);
            return;
        }
        if (s16) HUPR_LAUNCH(hupr_k_splitk_reduce4<16>, dim3((n4 + 15) / 16), dim3(256), 0, s, part, out, n4, splits, split_stride, taps, ci, out2, n_first);
        else HUPR_LAUNCH(hupr_k_splitk_reduce4<4>, dim3((n4 + 63) / 64), dim3(256), 0, s, part, out, n4, splits, split_stride, taps, ci, out2, n_first);
        return;
    }
    HUPR_LAUNCH(hupr_k_splitk_reduce, dim3((n + 255) / 256), dim3(256), 0, s, part, out, n, splits, split_stride,
                       taps, ci);
}

}  // namespace hupr

using namespace hupr;

// ---- C ABI ------------------------------------------------------------------------------------

extern "C" int hupr_gemm_f32(int ta, int tb, const float* A, const float* B, float* C, int M, int N,
                             int K, long lda, long ldb, long ldc, int batch, long a_bs, long b_bs,
                             long c_bs, const float* res, long res_ld, long res_bs, int accumulate,
                             hupr_stream_t stream) {
    HUPR_REQUIRE(A && B && C, "hupr_gemm_f32: null pointer");
    HUPR_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0, "hupr_gemm_f32: bad shape %d %d %d x%d", M, N, K, batch);
    HUPR_REQUIRE(batch <= 65535, "hupr_gemm_f32: batch %d > 65535", batch);
    GemmArgs a;
    fill_common(a);
    a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.a_bs0 = a_bs; a.b_bs0 = b_bs; a.c_bs0 = c_bs;
    a.res = res; a.res_ld = res_ld; a.res_bs0 = res_bs;
    a.accumulate = accumulate;
    hipStream_t s = as_stream(stream);
    if (ta == 0 && tb == 1) dispatch_tiles<A_ROWK, B_NK>(a, batch, s);        // C = A B^T
    else if (ta == 0 && tb == 0) dispatch_tiles<A_ROWK, B_KN>(a, batch, s);   // C = A B
    else if (ta == 1 && tb == 0) dispatch_tiles<A_KM, B_KN>(a, batch, s);     // C = A^T B
    else return fail(HUPR_ERR_ARG, "hupr_gemm_f32: unsupported transpose combination %d %d", ta, tb);
    HUPR_LAUNCH_OK("hupr_k_gemm_f32");
    return HUPR_OK;
}

// y[b,od,oh,ow, 0:Co] (row stride out_ld) = conv(x[b,:,:,:, 0:Ci] (voxel stride in_ld), wp[Co][taps][Ci])
//                                          (+ bias[Co]) (+ res[..., 0:Co] (row stride res_ld))
extern "C" int hupr_conv_fwd_f32(const float* x, const float* wp, const float* bias, const float* res,
                                 float* y, int Bn, int Di, int Hi, int Wi, int Ci, int in_ld, int Do,
                                 int Ho, int Wo, int Co, int out_ld, int res_ld, int kd, int kh, int kw,
                                 int pd, int ph, int pw, int accumulate, hupr_stream_t stream) {
    HUPR_REQUIRE(x && wp && y, "hupr_conv_fwd_f32: null pointer");
    GemmArgs a;
    fill_common(a);
    a.g = ConvGeom{Di, Hi, Wi, Ci, in_ld, Do, Ho, Wo, kd, kh, kw, pd, ph, pw};
    int rc = check_geom("hupr_conv_fwd_f32", Bn, a.g, Co, 32);
    if (rc) return rc;
    HUPR_REQUIRE(Do == Di + 2 * pd - kd + 1 && Ho == Hi + 2 * ph - kh + 1 && Wo == Wi + 2 * pw - kw + 1,
                 "hupr_conv_fwd_f32: output extent does not match stride-1 convolution");
    const long M = (long)Bn * Do * Ho * Wo;
    HUPR_REQUIRE(M < (1L << 31), "hupr_conv_fwd_f32: too many output voxels");
    a.A = x; a.B = wp; a.C = y;
    a.M = (int)M; a.N = Co; a.K = kd * kh * kw * Ci;
    a.lda = 0; a.ldb = a.K; a.ldc = out_ld;
    a.bias = bias; a.res = res; a.res_ld = res_ld;
    a.accumulate = accumulate;
    dispatch_tiles<A_CONV, B_NK>(a, 1, as_stream(stream));
    HUPR_LAUNCH_OK("hupr_k_gemm_f32<conv>");
    return HUPR_OK;
}

extern "C" size_t hupr_conv_wgrad_ws_bytes(int Bn, int Do, int Ho, int Wo, int Ci, int Co, int kd, int kh,
                                           int kw) {
    // number of voxel-axis slices (see wgrad_splits; the bf16 engine's 64-deep K tiles give the smaller tile count,
    // the fp32 engine's 32-deep ones the larger: take the larger slice count) x one partial weight tensor
    const size_t one = (size_t)Co * kd * kh * kw * Ci * sizeof(float);
    const long Mv = (long)Bn * Do * Ho * Wo;
    const long tiles = (long)((Co + ((Co <= 64) ? 64 : 128) - 1) / ((Co <= 64) ? 64 : 128)) * (((long)kd * kh * kw * Ci + 127) / 128);
    return (size_t)wgrad_splits(tiles, (Mv + 31) / 32, one) * one;
}

// dw (Co, Ci, kd, kh, kw) = sum over output voxels of dy[m][co] * x[m shifted by tap][ci]
extern "C" int hupr_conv_wgrad_f32(const float* x, const float* dy, float* dw, int Bn, int Di, int Hi,
                                   int Wi, int Ci, int in_ld, int Do, int Ho, int Wo, int Co, int dy_ld,
                                   int kd, int kh, int kw, int pd, int ph, int pw, void* ws,
                                   size_t ws_bytes, hupr_stream_t stream) {
    HUPR_REQUIRE(x && dy && dw && ws, "hupr_conv_wgrad_f32: null pointer");
    GemmArgs a;
    fill_common(a);
    a.g = ConvGeom{Di, Hi, Wi, Ci, in_ld, Do, Ho, Wo, kd, kh, kw, pd, ph, pw};
    int rc = check_geom("hupr_conv_wgrad_f32", Bn, a.g, Co, 32);
    if (rc) return rc;
    const long Mv = (long)Bn * Do * Ho * Wo;
    HUPR_REQUIRE(Mv < (1L << 31), "hupr_conv_wgrad_f32: too many output voxels");
    const int taps = kd * kh * kw;
    a.A = dy; a.B = x; a.C = reinterpret_cast<float*>(ws);
    a.M = Co; a.N = taps * Ci; a.K = (int)Mv;
    a.lda = dy_ld; a.ldb = 0; a.ldc = a.N;
    // 128-wide n tiles (spanning taps when Ci < 128); slice the voxel axis so the grid fills the chip
    const int bn = 128;
    const int bm = (Co <= 64) ? 64 : 128;
    const long tiles = (long)((Co + bm - 1) / bm) * ((a.N + bn - 1) / bn);
    const int ktiles = (int)((Mv + BK - 1) / BK);
    a.split_stride = (long)a.M * a.N;
    int splits = wgrad_splits(tiles, ktiles, (size_t)a.split_stride * sizeof(float));
    while (splits > 1 && (size_t)splits * a.split_stride * sizeof(float) > ws_bytes) splits >>= 1;
    a.ksplit = splits;
    if (ws_bytes < (size_t)splits * a.split_stride * sizeof(float))
        return fail(HUPR_ERR_WORKSPACE, "hupr_conv_wgrad_f32: workspace %zu < %zu", ws_bytes,
                    (size_t)splits * a.split_stride * sizeof(float));
    hipStream_t s = as_stream(stream);
    if (bm == 64) launch<64, 128, 1, 4, A_KM, B_CONVK>(a, 1, s);
    else launch<128, 128, 2, 2, A_KM, B_CONVK>(a, 1, s);
    HUPR_LAUNCH_OK("hupr_k_gemm_f32<wgrad>");
    launch_splitk_reduce(reinterpret_cast<const float*>(ws), dw, a.split_stride, splits, a.split_stride, taps, Ci, s);
    HUPR_LAUNCH_OK("hupr_k_splitk_reduce");
    return HUPR_OK;
}

extern "C" int hupr_pack_conv_weights_f32(const float* w, float* wp, int Co, int Ci, int taps, int mode,
                                          hupr_stream_t stream) {
    HUPR_REQUIRE(w && wp && Co > 0 && Ci > 0 && taps > 0 && (mode == 0 || mode == 1),
                 "hupr_pack_conv_weights_f32: bad argument");
    const long n = (long)Co * Ci * taps;
    HUPR_LAUNCH(hupr_k_pack_weights, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), w, wp,
                       Co, Ci, taps, mode);
    HUPR_LAUNCH_OK("hupr_k_pack_weights");
    return HUPR_OK;
}
