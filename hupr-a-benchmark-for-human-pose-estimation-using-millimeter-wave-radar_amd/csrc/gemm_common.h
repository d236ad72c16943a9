// Shared argument structures of the GEMM / implicit-GEMM convolution engines (fp32 and bf16 MFMA).
#pragma once
#include "hupr_common.h"

namespace hupr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum AMode { A_ROWK = 0, A_CONV = 1, A_KM = 2 };
enum BMode { B_NK = 0, B_KN = 1, B_CONVK = 2 };

struct ConvGeom {
    int Di, Hi, Wi, Ci;       // input voxels / channels taken part in the GEMM
    int in_ld;                // floats between consecutive voxels of the input buffer
    int Do, Ho, Wo;           // output voxels
    int kd, kh, kw;           // taps
    int pd, ph, pw;           // zero padding
};

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    int M, N, K;
    long lda, ldb, ldc;
    // batch: z -> (z / zdiv, z % zdiv), pointer += z0 * bs0 + z1 * bs1
    int zdiv;
    long a_bs0, a_bs1, b_bs0, b_bs1, c_bs0, c_bs1;
    ConvGeom g;
    const float* bias;        // [N] or null
    const float* res;         // residual added in the epilogue, same indexing as C with res_ld
    long res_ld, res_bs0, res_bs1;
    int ksplit;               // >1: grid.y slices K, partial tiles go to C + slice*M*ldc... (see host)
    long split_stride;        // floats between partial outputs
    int accumulate;           // 1: C += result (read-modify-write; not with ksplit)
    int flags;                // bf16 engine only: GEMM_A_BF16 / GEMM_B_BF16 / GEMM_C_BF16 (tensor stored as bf16 in HBM)
};
enum { GEMM_A_BF16 = 1, GEMM_B_BF16 = 2, GEMM_C_BF16 = 4 };


inline void fill_common(GemmArgs& a) {
    a.zdiv = 1;
    a.a_bs0 = a.a_bs1 = a.b_bs0 = a.b_bs1 = a.c_bs0 = a.c_bs1 = 0;
    a.bias = nullptr;
    a.res = nullptr;
    a.res_ld = a.res_bs0 = a.res_bs1 = 0;
    a.ksplit = 1;
    a.split_stride = 0;
    a.accumulate = 0;
    a.flags = 0;
    a.g = ConvGeom{};
}

inline int check_geom(const char* who, int Bn, const ConvGeom& g, int Co, int ci_mult) {
    HUPR_REQUIRE(Bn > 0 && g.Di > 0 && g.Hi > 0 && g.Wi > 0 && g.Do > 0 && g.Ho > 0 && g.Wo > 0,
                 "%s: bad geometry", who);
    HUPR_REQUIRE(g.Ci % ci_mult == 0, "%s: Cin=%d must be a multiple of %d", who, g.Ci, ci_mult);
    HUPR_REQUIRE(g.Do < 1024 && g.Ho < 1024 && g.Wo < 1024, "%s: extent >= 1024", who);
    HUPR_REQUIRE(g.in_ld % 4 == 0, "%s: input voxel stride %d not a multiple of 4 floats", who, g.in_ld);
    HUPR_REQUIRE(Co > 0, "%s: Cout=%d", who, Co);
    return HUPR_OK;
}

// split-K partial reduction + optional [Co][taps][Ci] -> (Co,Ci,taps) relayout (gemm_f32.hip)
// Voxel-axis slices of the GEMM-form weight gradient: enough workgroups for ~4 per CU, at most 256 slices and at most
// 256 MiB of partials (a 64x64 1x1 weight over 131k voxels used to run on 64 workgroups: a quarter of the chip).
// in_bytes (optional): bytes of the two operands.  Every slice writes one partial tensor that the reduction reads back, so the
// slices are also capped where that round trip would exceed the operands themselves (a 1024 x 256 gradient over 8 192 voxels
// ran 64 slices: 134 MB of partial traffic against 42 MB of input) — but never below one workgroup per CU.
inline int wgrad_splits(long tiles, long ktiles, size_t one_bytes, size_t in_bytes = 0) {
    long s = (1024 + tiles - 1) / tiles;
    if (in_bytes) {
        const long cap = (long)(in_bytes / (2 * one_bytes));
        const long floor_ = (256 + tiles - 1) / tiles;
        if (s > cap) s = cap > floor_ ? cap : floor_;
    }
    if (s > 256) s = 256;
    if (s > ktiles) s = ktiles;
    while (s > 1 && (size_t)s * one_bytes > ((size_t)256 << 20)) s >>= 1;
    return (int)(s < 1 ? 1 : s);
}
void launch_splitk_reduce(const float* part, float* out, long n, int splits, long split_stride, int taps, int ci,
                          hipStream_t s, float* out2 = nullptr, long n_first = 0);

}  // namespace hupr
