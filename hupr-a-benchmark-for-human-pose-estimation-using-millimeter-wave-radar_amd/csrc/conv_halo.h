// Shared argument block of the halo-tiled 3x3(x3) convolution kernels (conv_halo_bf16.hip, conv_halo256m_bf16.hip).
#pragma once
#include "gemm_common.h"

namespace hupr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // native 16-byte vector: stays in registers (a struct uint4 array did not)
typedef float f32x4n __attribute__((ext_vector_type(4)));

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// Activation storage of x / res / y: fp32 (ABF = false) or bf16 (ABF = true); leading dimensions are in elements.
struct HaloArgs {
    const void* x;           // [Bn][D][H][W] voxels, in_ld elements apart, Ci channels used
    const __bf16* wp;        // packed bf16 weights [Co][T][Ci]
    const float* bias;       // [Co] or null
    const void* res;         // residual (voxel stride res_ld) or null
    void* y;                 // [Bn][D][H][W] voxels, out_ld elements apart
    int Bn, D, H, W, Ci, in_ld, Co, out_ld, res_ld;
    int kd;                  // 1 or 3 (kh = kw = 3)
    int TD, log2TW;          // tile: TD x 8 x (1 << log2TW), TD * 8 * TW == 128
    int nd, nh, nw;          // tiles per axis
    int n_co_tiles;
    int ablate;              // profiling only: bit0 skip halo fill, bit1 skip MFMA stages, bit2 skip epilogue stores
    unsigned long long* trace;  // profiling only: s_memtime stamps of workgroup 0 / wave 0 (scripts/halo_trace.py) or null
    double* stats;           // 256-voxel kernel, bf16 output, no bias / residual: per-workgroup column sums [grid][2][Co] of the
                             // stored (rounded) output and its square — the BatchNorm statistics pass fused into the epilogue
    // 128-voxel kernel, small grids (single-sample inference): the (channel chunk, kz plane) units of the reduction are split over
    // blockIdx.y, each slice leaves fp32 partial sums [slice][voxel][Co] in `part` and hupr_k_conv_partial_reduce finishes
    // (slice order, bias, residual, one rounding).  part == nullptr: the whole reduction in one workgroup, as always.
    // (Finishing inside the kernel — the last slice to arrive sums the others — was built and measured: the slices of a tile
    // run on different XCDs with private L2s, and both ways of making the partial sums visible across them cost far more than the
    // 4 us launch they save: agent-scope scalar accesses 12.7 -> 42 us per layer, L2 write-back fences 94 us.)
    float* part;
    int units_per_slice;
    int no_res_prefetch;     // A/B aid (hupr_debug_halo_res_prefetch(0)): 256-voxel 16 x 16 x 32 kernel, residual read in the immediate epilogue as in rounds 4-5a
};

// fp32 partial sums of a K slice: this lane's voxel, its four 4-channel runs (see halo_store_voxel for the lane -> channel map)
__device__ __forceinline__ void halo_store_partial(const HaloArgs& p, const f32x16& acc, long m, int chb, int slice) {
    const long M = (long)p.Bn * p.D * p.H * p.W;
    float* dst = p.part + ((long)slice * M + m) * p.Co;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int ch = chb + 8 * g;
        if (ch < p.Co) *reinterpret_cast<f32x4n*>(dst + ch) = (f32x4n){acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    }
}

// 256-voxel persistent variant; returns false when the geometry is not supported
bool launch_conv_halo256(HaloArgs a, int Bn, bool abf, hipStream_t s);
bool conv_halo256_supported(const HaloArgs& a, int Bn, bool abf);
bool conv_halo256_stats_ok(const HaloArgs& a, int Bn);                         // fused BatchNorm statistics available for this launch?
void set_halo_tiles(int mask);                                                // test aid: which tiles of the 256-voxel kernel are in use (conv_halo256m_bf16.hip)
constexpr int kHalo256Grid = 256;      // persistent workgroups (= partial rows of HaloArgs::stats)

// Epilogue of both kernels.  The MFMAs are issued as D' = W * X^T, so a lane holds ONE voxel (column lane & 31 of the
// 32-voxel group) and 16 channels n = 8 g + 4 (lane >> 5) + j  (g = r >> 2, j = r & 3): four 4-channel runs that go
// out as 16-byte (fp32) or 8-byte (bf16) vector stores with a single voxel address per accumulator tile.
template <bool ABF>
__device__ __forceinline__ void halo_store_voxel(const HaloArgs& p, const f32x16& acc, long m, int chb) {
    if constexpr (ABF) {
        // bf16 output: lanes l and l + 32 hold the two 4-channel halves of the same voxel's 8-channel groups; one
        // v_permlane32_swap per dword gives each of them a full 16-byte run (two stores per tile instead of four 8-byte
        // ones — the epilogue is store-issue bound).  A residual is added to the lane's own channels before the swap.
        if ((p.Co & 7) == 0 && (p.out_ld & 7) == 0) {
            const int lh = (chb >> 2) & 1, cb = chb - 4 * lh;
            unsigned pk[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4n v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                const int ch = chb + 8 * g;
                if (p.bias && ch < p.Co) v += (f32x4n){p.bias[ch], p.bias[ch + 1], p.bias[ch + 2], p.bias[ch + 3]};
                if (p.res && ch < p.Co) {
                    const bf16x4 rv = *reinterpret_cast<const bf16x4*>(static_cast<const __bf16*>(p.res) + m * p.res_ld + ch);
                    v += (f32x4n){(float)rv[0], (float)rv[1], (float)rv[2], (float)rv[3]};
                }
                typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                const bf16x2 lo = {(__bf16)v[0], (__bf16)v[1]}, hi = {(__bf16)v[2], (__bf16)v[3]};
                pk[g][0] = __builtin_bit_cast(unsigned, lo);
                pk[g][1] = __builtin_bit_cast(unsigned, hi);
            }
#pragma unroll
            for (int gp = 0; gp < 4; gp += 2) {
                const auto r0 = __builtin_amdgcn_permlane32_swap(pk[gp][0], pk[gp + 1][0], false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(pk[gp][1], pk[gp + 1][1], false, false);
                const int ch = cb + 8 * (gp + lh);            // lower half-wave: group gp, upper: group gp + 1
                if (ch < p.Co)
                    *reinterpret_cast<u32x4*>(static_cast<__bf16*>(p.y) + m * p.out_ld + ch) = (u32x4){r0[0], r1[0], r0[1], r1[1]};
            }
            return;
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int ch = chb + 8 * g;
        if (ch < p.Co) {
            f32x4n v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            if (p.bias) v += (f32x4n){p.bias[ch], p.bias[ch + 1], p.bias[ch + 2], p.bias[ch + 3]};   // parameters may be 4-byte aligned views
            if constexpr (ABF) {
                if (p.res) {
                    const bf16x4 rv = *reinterpret_cast<const bf16x4*>(static_cast<const __bf16*>(p.res) + m * p.res_ld + ch);
                    v += (f32x4n){(float)rv[0], (float)rv[1], (float)rv[2], (float)rv[3]};
                }
                const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                *reinterpret_cast<bf16x4*>(static_cast<__bf16*>(p.y) + m * p.out_ld + ch) = o;
            } else {
                if (p.res) v += *reinterpret_cast<const f32x4n*>(static_cast<const float*>(p.res) + m * p.res_ld + ch);
                *reinterpret_cast<f32x4n*>(static_cast<float*>(p.y) + m * p.out_ld + ch) = v;
            }
        }
    }
}

// Deferred bf16 epilogue of the 256-voxel kernel (no bias / residual): a finished accumulator tile is parked as 8 packed
// bf16 pairs (halo_pack_tile: pk[2 g] / pk[2 g + 1] = channels 0-1 / 2-3 of group g) and stored in two pieces — channel groups
// gp, gp + 1 (gp = 0 or 2): two permlane32_swap + one 16-byte store — between the MFMA groups of the NEXT tile's first stage,
// where they cost no matrix-pipe time.
__device__ __forceinline__ void halo_pack_tile(const f32x16& acc, unsigned* pk) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const bf16x2 lo = {(__bf16)acc[4 * g], (__bf16)acc[4 * g + 1]};
        const bf16x2 hi = {(__bf16)acc[4 * g + 2], (__bf16)acc[4 * g + 3]};
        pk[2 * g] = __builtin_bit_cast(unsigned, lo);
        pk[2 * g + 1] = __builtin_bit_cast(unsigned, hi);
    }
}
__device__ __forceinline__ void halo_store_packed_part(const HaloArgs& p, const unsigned* pk, long m, int chb, int gp) {
    const int lh = (chb >> 2) & 1, cb = chb - 4 * lh;
    const auto r0 = __builtin_amdgcn_permlane32_swap(pk[2 * gp], pk[2 * gp + 2], false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(pk[2 * gp + 1], pk[2 * gp + 3], false, false);
    const int ch = cb + 8 * (gp + lh);
    *reinterpret_cast<u32x4*>(static_cast<__bf16*>(p.y) + m * p.out_ld + ch) = (u32x4){r0[0], r1[0], r0[1], r1[1]};
}

}  // namespace hupr
