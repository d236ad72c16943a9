// bf16-MFMA GEMM / implicit-GEMM convolution engine for gfx950 (fp32 in HBM, fp32 accumulate).
//
// Same operand modes, batching, epilogue and C ABI shape as gemm_f32.hip, but the matrix pipe runs
// v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate).  Tensors stay fp32 in HBM; operands are
// rounded to bf16 (RNE, v_cvt_pk_bf16_f32) on their way into LDS, so nothing else in the pipeline
// changes and the fp32 engine remains the parity reference.
//
// Tiling: 256 threads = 4 waves, block tile BM x BN x 64; LDS images are k-contiguous bf16 rows of
// 64 elements padded to 144 B so that the ds_read_b128 fragment reads (lane = row, 8 consecutive
// k per lane) are bank-conflict free; staging stores are 16-B (or 8-B) writes to contiguous runs.
// k-major operands (A^T, B, im2col for the weight gradient) are transposed in registers: a thread
// loads eight float4 along the contiguous axis for eight consecutive k and emits four 16-B rows.
#include "gemm_common.h"

namespace hupr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4n __attribute__((ext_vector_type(4)));

constexpr int BKH = 64;                 // k elements per tile
constexpr int LDH = BKH + 8;            // LDS row stride in bf16 elements (144 bytes)

__device__ __forceinline__ bf16x8 cvt8(const float4& a, const float4& b) {
    bf16x8 r;
    r[0] = (__bf16)a.x; r[1] = (__bf16)a.y; r[2] = (__bf16)a.z; r[3] = (__bf16)a.w;
    r[4] = (__bf16)b.x; r[5] = (__bf16)b.y; r[6] = (__bf16)b.z; r[7] = (__bf16)b.w;
    return r;
}

// guarded float4 load along a contiguous axis: elements [i, i+4) of a run of length n
__device__ __forceinline__ float4 ld4(const float* __restrict__ src, int i, int n, bool vec_ok) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i + 3 < n && vec_ok) {
        const f32x4n t = *reinterpret_cast<const f32x4n*>(src);      // native vector: stays one dwordx4 load
        v = make_float4(t.x, t.y, t.z, t.w);
    } else {
        if (i < n) v.x = src[0];
        if (i + 1 < n) v.y = src[1];
        if (i + 2 < n) v.z = src[2];
        if (i + 3 < n) v.w = src[3];
    }
    return v;
}

// eight / four bf16 values stored in HBM -> fp32 registers (exact: a bf16 is the top half of an fp32)
__device__ __forceinline__ void ld8_bf16(const void* src, float4& v0, float4& v1) {
    const uint4 t = *reinterpret_cast<const uint4*>(src);
    v0 = make_float4(__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16),
                     __uint_as_float(t.y & 0xffff0000u));
    v1 = make_float4(__uint_as_float(t.z << 16), __uint_as_float(t.z & 0xffff0000u), __uint_as_float(t.w << 16),
                     __uint_as_float(t.w & 0xffff0000u));
}
__device__ __forceinline__ float4 ld4_bf16(const void* src) {
    const uint2 t = *reinterpret_cast<const uint2*>(src);
    return make_float4(__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16),
                       __uint_as_float(t.y & 0xffff0000u));
}

struct VoxDec {
    int b, od, oh, ow;
};
__device__ __forceinline__ VoxDec decode_vox(int row, const ConvGeom& g) {
    VoxDec v;
    v.ow = row % g.Wo;
    int t = row / g.Wo;
    v.oh = t % g.Ho;
    t /= g.Ho;
    v.od = t % g.Do;
    v.b = t / g.Do;
    return v;
}

template <int BM, int BN, int WM, int WN, int AM, int BMD>
__global__ __launch_bounds__(256) void hupr_k_gemm_bf16(GemmArgs p) {
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "bad tiling");
    static_assert(BM == 64 || BM == 128, "BM");
    static_assert(BN == 64 || BN == 128, "BN");
    // row-contiguous operands: chunk = 8 consecutive k; 8 chunks per row; 32 rows per pass
    constexpr int A_CH = BM / 32, B_CH = BN / 32;      // chunks (k-contiguous modes) per thread
    // k-major operands: thread = (c = t&7: k-chunk of KC rows, g = t>>3: 4 consecutive columns)
    constexpr int A_KC = (BM == 128) ? 8 : 4, B_KC = (BN == 128) ? 8 : 4;

    __shared__ __attribute__((aligned(16))) __bf16 As[BM * LDH];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[BN * LDH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
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

    const int k_tiles_total = (p.K + BKH - 1) / BKH;
    const int per = (k_tiles_total + p.ksplit - 1) / p.ksplit;
    const int kt_begin = blockIdx.y * per;
    const int kt_end = min(k_tiles_total, kt_begin + per);
    if (p.ksplit > 1) Cg += (long)blockIdx.y * p.split_stride;
    const ConvGeom& g = p.g;

    // ---- staging registers -------------------------------------------------------------------
    // k-contiguous: per chunk two float4 (8 k);   k-major: KC float4 (KC k rows x 4 columns)
    constexpr int A_NV = (AM == A_KM) ? A_KC : 2 * A_CH;
    constexpr int B_NV = (BMD == B_NK) ? 2 * B_CH : B_KC;
    float4 ra[A_NV], rb[B_NV];

    // per-thread constants of the k-contiguous A loaders
    const int ck = tid & 7;                 // chunk index inside the 64-wide k tile
    long a_off[A_CH];
    int a_vox[A_CH];
    if (AM == A_ROWK) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int row = m0 + (tid >> 3) + 32 * i;
            a_off[i] = (row < p.M) ? (long)row * p.lda : -1;
        }
    } else if (AM == A_CONV) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int row = m0 + (tid >> 3) + 32 * i;
            if (row < p.M) {
                const VoxDec v = decode_vox(row, g);
                a_vox[i] = (v.od << 20) | (v.oh << 10) | v.ow;
                a_off[i] = (long)v.b * g.Di * g.Hi * g.Wi;
            } else {
                a_vox[i] = -1;
                a_off[i] = 0;
            }
        }
    }
    const bool a_vec = (p.lda & 3) == 0, b_vec = (p.ldb & 3) == 0;
    // k-major operands: (k-chunk, 4-column group) owned by this thread.  Width 128: 8 chunks of 8 k x 32 groups;
    // width 64: 16 chunks of 4 k x 16 groups.
    const int a_kc = (BM == 128) ? (tid & 7) : ((tid & 7) + 8 * (tid >> 7));
    const int a_cg = (BM == 128) ? (tid >> 3) : ((tid >> 3) & 15);
    const int b_kc = (BN == 128) ? (tid & 7) : ((tid & 7) + 8 * (tid >> 7));
    const int b_cg = (BN == 128) ? (tid >> 3) : ((tid >> 3) & 15);

    auto load_a = [&](int kt) {
        const int k0 = kt * BKH;
        if (AM == A_ROWK) {
            const int k = k0 + ck * 8;
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
                if (a_off[i] >= 0) {
                    const float* src = Ag + a_off[i] + k;
                    v0 = ld4(src, k, p.K, a_vec);
                    v1 = ld4(src + 4, k + 4, p.K, a_vec);
                }
                ra[2 * i] = v0;
                ra[2 * i + 1] = v1;
            }
        } else if (AM == A_CONV) {
            const int k = k0 + ck * 8;                 // Ci % 8 == 0: a chunk never straddles taps
            const int tap = k / g.Ci, ci = k - tap * g.Ci;
            const int tw_ = tap % g.kw, tt = tap / g.kw;
            const int th_ = tt % g.kh, td_ = tt / g.kh;
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
                if (a_vox[i] >= 0 && k < p.K) {
                    const int id = (a_vox[i] >> 20) + td_ - g.pd;
                    const int ih = ((a_vox[i] >> 10) & 1023) + th_ - g.ph;
                    const int iw = (a_vox[i] & 1023) + tw_ - g.pw;
                    if ((unsigned)id < (unsigned)g.Di && (unsigned)ih < (unsigned)g.Hi && (unsigned)iw < (unsigned)g.Wi) {
                        const long e = (a_off[i] + ((long)id * g.Hi + ih) * g.Wi + iw) * g.in_ld + ci;
                        if (p.flags & GEMM_A_BF16) {
                            ld8_bf16(reinterpret_cast<const __bf16*>(Ag) + e, v0, v1);
                        } else {
                            v0 = *reinterpret_cast<const float4*>(Ag + e);
                            v1 = *reinterpret_cast<const float4*>(Ag + e + 4);
                        }
                    }
                }
                ra[2 * i] = v0;
                ra[2 * i + 1] = v1;
            }
        } else {   // A_KM: A[k][m]; thread: k rows k0 + a_kc*KC + j, columns m0 + 4*a_cg ..+3
            const int m = m0 + 4 * a_cg;
#pragma unroll
            for (int j = 0; j < A_KC; ++j) {
                const int k = k0 + a_kc * A_KC + j;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < p.K) v = ld4(Ag + (long)k * p.lda + m, m, p.M, a_vec);
                ra[j] = v;
            }
        }
    };

    auto load_b = [&](int kt) {
        const int k0 = kt * BKH;
        if (BMD == B_NK) {
            const int k = k0 + ck * 8;
#pragma unroll
            for (int i = 0; i < B_CH; ++i) {
                const int n = n0 + (tid >> 3) + 32 * i;
                float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
                if (n < p.N) {
                    const float* src = Bg + (long)n * p.ldb + k;
                    v0 = ld4(src, k, p.K, b_vec);
                    v1 = ld4(src + 4, k + 4, p.K, b_vec);
                }
                rb[2 * i] = v0;
                rb[2 * i + 1] = v1;
            }
        } else if (BMD == B_KN) {
            const int n = n0 + 4 * b_cg;
#pragma unroll
            for (int j = 0; j < B_KC; ++j) {
                const int k = k0 + b_kc * B_KC + j;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < p.K) v = ld4(Bg + (long)k * p.ldb + n, n, p.N, b_vec);
                rb[j] = v;
            }
        } else {   // B_CONVK: k = output voxel, n = tap*Ci + ci (4 consecutive ci share a tap: Ci % 4 == 0)
            const int n = n0 + 4 * b_cg;
            const int tap = n / g.Ci, ci = n - tap * g.Ci;
            const int tw_ = tap % g.kw, tt = tap / g.kw;
            const int th_ = tt % g.kh, td_ = tt / g.kh;
#pragma unroll
            for (int j = 0; j < B_KC; ++j) {
                const int k = k0 + b_kc * B_KC + j;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < p.K && n < p.N) {
                    const VoxDec d = decode_vox(k, g);
                    const int id = d.od + td_ - g.pd, ih = d.oh + th_ - g.ph, iw = d.ow + tw_ - g.pw;
                    if ((unsigned)id < (unsigned)g.Di && (unsigned)ih < (unsigned)g.Hi && (unsigned)iw < (unsigned)g.Wi) {
                        const long e = ((((long)d.b * g.Di + id) * g.Hi + ih) * g.Wi + iw) * g.in_ld + ci;
                        if (p.flags & GEMM_B_BF16) v = ld4_bf16(reinterpret_cast<const __bf16*>(Bg) + e);
                        else v = *reinterpret_cast<const float4*>(Bg + e);
                    }
                }
                rb[j] = v;
            }
        }
    };

    // transposed store: KC k-rows x 4 columns in registers -> 4 LDS rows (one per column), KC bf16 each
    auto store_km = [&](__bf16* S, const float4* r, int KC, int kc, int cg) {
        const int col = 4 * cg;
        const int koff = kc * KC;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            __bf16* dst = S + (col + c) * LDH + koff;
            if (KC == 8) {
                bf16x8 v;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (__bf16)(c == 0 ? r[j].x : c == 1 ? r[j].y : c == 2 ? r[j].z : r[j].w);
                *reinterpret_cast<bf16x8*>(dst) = v;
            } else {
                bf16x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (__bf16)(c == 0 ? r[j].x : c == 1 ? r[j].y : c == 2 ? r[j].z : r[j].w);
                *reinterpret_cast<bf16x4*>(dst) = v;
            }
        }
    };

    auto store_tiles = [&]() {
        if (AM == A_KM) {
            store_km(As, ra, A_KC, a_kc, a_cg);
        } else {
#pragma unroll
            for (int i = 0; i < A_CH; ++i)
                *reinterpret_cast<bf16x8*>(&As[((tid >> 3) + 32 * i) * LDH + ck * 8]) = cvt8(ra[2 * i], ra[2 * i + 1]);
        }
        if (BMD == B_NK) {
#pragma unroll
            for (int i = 0; i < B_CH; ++i)
                *reinterpret_cast<bf16x8*>(&Bs[((tid >> 3) + 32 * i) * LDH + ck * 8]) = cvt8(rb[2 * i], rb[2 * i + 1]);
        } else {
            store_km(Bs, rb, B_KC, b_kc, b_cg);
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
        for (int ks = 0; ks < BKH / 16; ++ks) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = *reinterpret_cast<const bf16x8*>(&As[(wm * WTM + i * 32 + lr) * LDH + ks * 16 + lh * 8]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[j] = *reinterpret_cast<const bf16x8*>(&Bs[(wn * WTN + j * 32 + lr) * LDH + ks * 16 + lh * 8]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    // D' = B A^T: the accumulator tile comes out transposed — a lane owns ONE output row (i, lr) and four runs of
                    // four consecutive columns, so the epilogue stores 16-byte (fp32) / 8-byte (bf16) vectors instead of 16
                    // scalars per tile (the skinny, HBM-bound GEMMs of this path spent most of their time issuing 2-byte stores)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    const float* __restrict__ resg = p.res ? p.res + z0 * p.res_bs0 + z1 * p.res_bs1 : nullptr;
    typedef __bf16 bf16x4e __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = m0 + wm * WTM + i * 32 + lr;             // this lane's output row
        if (row >= p.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = n0 + wn * WTN + j * 32 + 8 * g + 4 * lh;      // first of four consecutive columns
                if (col >= p.N) continue;
                float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                const bool full = col + 3 < p.N;
                if (p.bias) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (full || col + k < p.N) v[k] += p.bias[col + k];
                }
                if (resg) {
                    const float* rp = resg + (long)row * p.res_ld + col;
                    if (full && (reinterpret_cast<uintptr_t>(rp) & 15) == 0) {
                        const float4 r4 = *reinterpret_cast<const float4*>(rp);
                        v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) if (full || col + k < p.N) v[k] += rp[k];
                    }
                }
                if (p.flags & GEMM_C_BF16) {            // bf16-stored output (batch strides in elements)
                    __bf16* dst = reinterpret_cast<__bf16*>(p.C) + z0 * p.c_bs0 + z1 * p.c_bs1 + (long)row * p.ldc + col;
                    if (full && (reinterpret_cast<uintptr_t>(dst) & 7) == 0) {
                        *reinterpret_cast<bf16x4e*>(dst) = (bf16x4e){(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) if (full || col + k < p.N) dst[k] = (__bf16)v[k];
                    }
                } else {
                    float* dst = Cg + (long)row * p.ldc + col;
                    if (full && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
                        if (p.accumulate) {
                            const float4 o = *reinterpret_cast<const float4*>(dst);
                            v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
                        }
                        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if (full || col + k < p.N) {
                                if (p.accumulate) v[k] += dst[k];
                                dst[k] = v[k];
                            }
                        }
                    }
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int AM, int BMD>
static void launch_h(const GemmArgs& a, int batch, hipStream_t s) {
    const int mt = (a.M + BM - 1) / BM, nt = (a.N + BN - 1) / BN;
    HUPR_LAUNCH((hupr_k_gemm_bf16<BM, BN, WM, WN, AM, BMD>), dim3(mt * nt, a.ksplit, batch), dim3(256), 0, s, a);
}

template <int AM, int BMD>
static void dispatch_h(const GemmArgs& a, int batch, hipStream_t s) {
    auto blocks = [&](int bm, int bn) { return (long)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn) * batch; };
    if (a.N > 64) {
        if (a.M > 64 && blocks(128, 128) >= 512) launch_h<128, 128, 2, 2, AM, BMD>(a, batch, s);
        else if (blocks(64, 128) >= 512) launch_h<64, 128, 1, 4, AM, BMD>(a, batch, s);
        else launch_h<64, 64, 2, 2, AM, BMD>(a, batch, s);      // small problems (level-0 attention GEMMs): two workgroups per CU
    } else {
        if (a.M > 64 && blocks(128, 64) >= 512) launch_h<128, 64, 2, 2, AM, BMD>(a, batch, s);
        else launch_h<64, 64, 2, 2, AM, BMD>(a, batch, s);
    }
}

}  // namespace hupr

using namespace hupr;

extern "C" int hupr_gemm_bf16(int ta, int tb, const float* A, const float* B, float* C, int M, int N, int K, long lda,
                              long ldb, long ldc, int batch, long a_bs, long b_bs, long c_bs, const float* res,
                              long res_ld, long res_bs, int accumulate, hupr_stream_t stream) {
    HUPR_REQUIRE(A && B && C, "hupr_gemm_bf16: null pointer");
    HUPR_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0 && batch <= 65535, "hupr_gemm_bf16: bad shape %d %d %d x%d", M, N, K, batch);
    GemmArgs a;
    fill_common(a);
    a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.a_bs0 = a_bs; a.b_bs0 = b_bs; a.c_bs0 = c_bs;
    a.res = res; a.res_ld = res_ld; a.res_bs0 = res_bs;
    a.accumulate = accumulate;
    hipStream_t s = as_stream(stream);
    if (ta == 0 && tb == 1) dispatch_h<A_ROWK, B_NK>(a, batch, s);
    else if (ta == 0 && tb == 0) dispatch_h<A_ROWK, B_KN>(a, batch, s);
    else if (ta == 1 && tb == 0) dispatch_h<A_KM, B_KN>(a, batch, s);
    else return fail(HUPR_ERR_ARG, "hupr_gemm_bf16: unsupported transpose combination %d %d", ta, tb);
    HUPR_LAUNCH_OK("hupr_k_gemm_bf16");
    return HUPR_OK;
}

static int conv_fwd_h(const char* who, const void* x, const float* wp, const float* bias, const float* res, void* y,
                      int Bn, int Di, int Hi, int Wi, int Ci, int in_ld, int Do, int Ho, int Wo, int Co, int out_ld,
                      int res_ld, int kd, int kh, int kw, int pd, int ph, int pw, int accumulate, int flags,
                      hupr_stream_t stream) {
    HUPR_REQUIRE(x && wp && y, "%s: null pointer", who);
    GemmArgs a;
    fill_common(a);
    a.g = ConvGeom{Di, Hi, Wi, Ci, in_ld, Do, Ho, Wo, kd, kh, kw, pd, ph, pw};
    int rc = check_geom(who, Bn, a.g, Co, 8);
    if (rc) return rc;
    HUPR_REQUIRE(Do == Di + 2 * pd - kd + 1 && Ho == Hi + 2 * ph - kh + 1 && Wo == Wi + 2 * pw - kw + 1,
                 "%s: output extent does not match stride-1 convolution", who);
    HUPR_REQUIRE(!(flags & GEMM_C_BF16) || (!accumulate && !res), "%s: bf16 output excludes accumulate/residual", who);
    const long M = (long)Bn * Do * Ho * Wo;
    HUPR_REQUIRE(M < (1L << 31), "%s: too many output voxels", who);
    a.A = reinterpret_cast<const float*>(x); a.B = wp; a.C = reinterpret_cast<float*>(y);
    a.M = (int)M; a.N = Co; a.K = kd * kh * kw * Ci;
    a.lda = 0; a.ldb = a.K; a.ldc = out_ld;
    a.bias = bias; a.res = res; a.res_ld = res_ld;
    a.accumulate = accumulate;
    a.flags = flags;
    dispatch_h<A_CONV, B_NK>(a, 1, as_stream(stream));
    HUPR_LAUNCH_OK("hupr_k_gemm_bf16<conv>");
    return HUPR_OK;
}

extern "C" int hupr_conv_fwd_bf16(const float* x, const float* wp, const float* bias, const float* res, float* y, int Bn,
                                  int Di, int Hi, int Wi, int Ci, int in_ld, int Do, int Ho, int Wo, int Co, int out_ld,
                                  int res_ld, int kd, int kh, int kw, int pd, int ph, int pw, int accumulate,
                                  hupr_stream_t stream) {
    return conv_fwd_h("hupr_conv_fwd_bf16", x, wp, bias, res, y, Bn, Di, Hi, Wi, Ci, in_ld, Do, Ho, Wo, Co, out_ld, res_ld,
                      kd, kh, kw, pd, ph, pw, accumulate, 0, stream);
}

extern "C" int hupr_conv_fwd_bf16_mixed(const void* x, int x_bf16, const float* wp, const float* bias, void* y, int y_bf16,
                                        int Bn, int Di, int Hi, int Wi, int Ci, int in_ld, int Do, int Ho, int Wo, int Co,
                                        int out_ld, int kd, int kh, int kw, int pd, int ph, int pw, hupr_stream_t stream) {
    return conv_fwd_h("hupr_conv_fwd_bf16_mixed", x, wp, bias, nullptr, y, Bn, Di, Hi, Wi, Ci, in_ld, Do, Ho, Wo, Co, out_ld,
                      0, kd, kh, kw, pd, ph, pw, 0, (x_bf16 ? GEMM_A_BF16 : 0) | (y_bf16 ? GEMM_C_BF16 : 0), stream);
}

// Input gradient of a temporal merge (kernel (G,1,1), no padding, one output slice): every input slice d sees exactly
// one tap, so instead of an implicit GEMM over G zero-padded taps this is G*Bn plain GEMMs
// dx[b, d] (HW x Ci) = dy[b] (HW x Co) . W_d (Co x Ci), batched over z = (b, d).  wp1: mode-1 packed weights
// [Ci][taps reversed][Co] (fp32); dx is stored as bf16 when dx_bf16.
extern "C" int hupr_tmerge_dgrad_bf16(const float* dy, const float* wp1, void* dx, int dx_bf16, int Bn, int G, int HW,
                                      int Ci, int Co, hupr_stream_t stream) {
    HUPR_REQUIRE(dy && wp1 && dx, "hupr_tmerge_dgrad_bf16: null pointer");
    HUPR_REQUIRE(Bn > 0 && G > 0 && HW > 0 && Ci > 0 && Co > 0 && (long)Bn * G <= 65535,
                 "hupr_tmerge_dgrad_bf16: bad shape %d %d %d %d %d", Bn, G, HW, Ci, Co);
    GemmArgs a;
    fill_common(a);
    a.A = dy; a.B = wp1 + (long)(G - 1) * Co; a.C = reinterpret_cast<float*>(dx);
    a.M = HW; a.N = Ci; a.K = Co;
    a.lda = Co; a.ldb = (long)G * Co; a.ldc = Ci;
    a.zdiv = G;
    a.a_bs0 = (long)HW * Co; a.a_bs1 = 0;
    a.b_bs0 = 0; a.b_bs1 = -(long)Co;
    a.c_bs0 = (long)G * HW * Ci; a.c_bs1 = (long)HW * Ci;
    a.flags = dx_bf16 ? GEMM_C_BF16 : 0;
    dispatch_h<A_ROWK, B_NK>(a, Bn * G, as_stream(stream));
    HUPR_LAUNCH_OK("hupr_k_gemm_bf16<tmerge dgrad>");
    return HUPR_OK;
}

static int conv_wgrad_h(const void* x, int x_bf16, const float* dy, float* dw, int Bn, int Di, int Hi, int Wi, int Ci,
                        int in_ld, int Do, int Ho, int Wo, int Co, int dy_ld, int kd, int kh, int kw, int pd,
                        int ph, int pw, void* ws, size_t ws_bytes, hupr_stream_t stream) {
    HUPR_REQUIRE(x && dy && dw && ws, "hupr_conv_wgrad_bf16: null pointer");
    GemmArgs a;
    fill_common(a);
    a.g = ConvGeom{Di, Hi, Wi, Ci, in_ld, Do, Ho, Wo, kd, kh, kw, pd, ph, pw};
    int rc = check_geom("hupr_conv_wgrad_bf16", Bn, a.g, Co, 4);
    if (rc) return rc;
    const long Mv = (long)Bn * Do * Ho * Wo;
    HUPR_REQUIRE(Mv < (1L << 31), "hupr_conv_wgrad_bf16: too many output voxels");
    const int taps = kd * kh * kw;
    a.A = dy; a.B = reinterpret_cast<const float*>(x); a.C = reinterpret_cast<float*>(ws);
    a.flags = x_bf16 ? GEMM_B_BF16 : 0;
    a.M = Co; a.N = taps * Ci; a.K = (int)Mv;
    a.lda = dy_ld; a.ldb = 0; a.ldc = a.N;
    const int bm = (Co <= 64) ? 64 : 128;
    const long tiles = (long)((Co + bm - 1) / bm) * ((a.N + 127) / 128);
    const int ktiles = (int)((Mv + BKH - 1) / BKH);
    a.split_stride = (long)a.M * a.N;
    const size_t in_bytes = (size_t)Mv * ((size_t)Co * sizeof(float) + (size_t)taps * Ci * (x_bf16 ? 2 : 4));
    int splits = wgrad_splits(tiles, ktiles, (size_t)a.split_stride * sizeof(float), in_bytes);
    while (splits > 1 && (size_t)splits * a.split_stride * sizeof(float) > ws_bytes) splits >>= 1;
    a.ksplit = splits;
    if (ws_bytes < (size_t)splits * a.split_stride * sizeof(float))
        return fail(HUPR_ERR_WORKSPACE, "hupr_conv_wgrad_bf16: workspace %zu < %zu", ws_bytes,
                    (size_t)splits * a.split_stride * sizeof(float));
    hipStream_t s = as_stream(stream);
    if (bm == 64) launch_h<64, 128, 1, 4, A_KM, B_CONVK>(a, 1, s);
    else launch_h<128, 128, 2, 2, A_KM, B_CONVK>(a, 1, s);
    HUPR_LAUNCH_OK("hupr_k_gemm_bf16<wgrad>");
    launch_splitk_reduce(reinterpret_cast<const float*>(ws), dw, a.split_stride, splits, a.split_stride, taps, Ci, s);
    HUPR_LAUNCH_OK("hupr_k_splitk_reduce");
    return HUPR_OK;
}

extern "C" int hupr_conv_wgrad_bf16(const float* x, const float* dy, float* dw, int Bn, int Di, int Hi, int Wi, int Ci,
                                    int in_ld, int Do, int Ho, int Wo, int Co, int dy_ld, int kd, int kh, int kw, int pd,
                                    int ph, int pw, void* ws, size_t ws_bytes, hupr_stream_t stream) {
    return conv_wgrad_h(x, 0, dy, dw, Bn, Di, Hi, Wi, Ci, in_ld, Do, Ho, Wo, Co, dy_ld, kd, kh, kw, pd, ph, pw, ws, ws_bytes,
                        stream);
}

extern "C" int hupr_conv_wgrad_bf16_mixed(const void* x, int x_bf16, const float* dy, float* dw, int Bn, int Di, int Hi,
                                          int Wi, int Ci, int in_ld, int Do, int Ho, int Wo, int Co, int dy_ld, int kd,
                                          int kh, int kw, int pd, int ph, int pw, void* ws, size_t ws_bytes,
                                          hupr_stream_t stream) {
    return conv_wgrad_h(x, x_bf16, dy, dw, Bn, Di, Hi, Wi, Ci, in_ld, Do, Ho, Wo, Co, dy_ld, kd, kh, kw, pd, ph, pw, ws,
                        ws_bytes, stream);
}
