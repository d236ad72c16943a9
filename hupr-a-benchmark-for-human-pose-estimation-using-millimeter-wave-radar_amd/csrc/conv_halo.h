// Shared argument block of the halo-tiled 3x3(x3) convolution kernels (conv_halo_bf16.hip, conv_halo256_bf16.hip).
#pragma once
#include "gemm_common.h"

namespace hupr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // native 16-byte vector: stays in registers (a struct uint4 array did not)
typedef float f32x4n __attribute__((ext_vector_type(4)));

struct HaloArgs {
    const float* x;          // [Bn][D][H][W] voxels, in_ld floats apart, Ci channels used
    const __bf16* wp;        // packed bf16 weights [Co][T][Ci]
    const float* bias;       // [Co] or null
    const float* res;        // residual (voxel stride res_ld) or null
    float* y;                // [Bn][D][H][W] voxels, out_ld floats apart
    int Bn, D, H, W, Ci, in_ld, Co, out_ld, res_ld;
    int kd;                  // 1 or 3 (kh = kw = 3)
    int TD, log2TW;          // tile: TD x 8 x (1 << log2TW), TD * 8 * TW == 128
    int nd, nh, nw;          // tiles per axis
    int n_co_tiles;
    int ablate;              // profiling only: bit0 skip halo fill, bit1 skip MFMA stages, bit2 skip epilogue stores
};

// 256-voxel persistent variant; returns false when the geometry is not supported
bool launch_conv_halo256(HaloArgs a, int Bn, hipStream_t s);

}  // namespace hupr
