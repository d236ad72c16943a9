// LDS halo-tiled weight gradient of the 3x3x3 / 1x3x3 "same" convolutions on the bf16 matrix pipe.
//
//   dW[co][tap][ci] = sum over output voxels v of dy[v][co] * x[v + tap][ci]
//
// Both MFMA operands need the reduction axis (voxels) along K, i.e. the transposes of the row-major
// channels-last images.  gfx950's ds_read_b64_tr_b16 does that transpose on the way out of LDS: inside
// each 16-lane group, lane l receives element (l & 3) of the 8-byte segments addressed by lanes
// (l >> 2) + 4j (j = 0..3).  If supplier lane s points at row rho[s >> 2], columns 4*(s & 3).., lane l gets
// column l of four freely chosen rows — so the x fragment for ANY tap shift comes straight from one
// row-major halo image (row address = halo voxel + tap offset; no alignment constraint), and the dy
// fragment from the row-major dy tile.  (Semantics pinned on hardware by scripts/probes/tr_probe.hip.)
//
// A workgroup owns one depth tap plane td (9 taps), a 64x64 (co x ci) weight tile, and walks a strided
// set of 128-voxel spatial tiles, keeping all 9 x (64x64) partial sums in registers (144 acc regs per
// lane); per tile it stages the dy tile and the td-plane x halo once (fp32 -> bf16).  Global bytes per
// flop drop ~5x against the implicit-GEMM weight gradient.  Partials [group][Co][T][Ci] are reduced by
// the deterministic split-K kernel, which also emits the parameter layout (Co,Ci,kd,kh,kw).
#include "gemm_common.h"

namespace hupr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct WgradHaloArgs {
    const void* x;         // [Bn][D][H][W] voxels, in_ld elements apart   (fp32, or bf16 when ABF)
    const void* dy;        // [Bn][D][H][W] voxels, dy_ld elements apart
    float* part;           // [groups][Co][T][Ci]
    int Bn, D, H, W, Ci, in_ld, Co, dy_ld;
    int kd, TD, log2TW, nd, nh, nw;
    int n_ci_tiles, n_co_tiles, groups, n_spatial;
    int xcd_map;           // LDS-DMA kernel: 1-D XCD-aware grid (see hupr_k_wgrad_halo_glds)
    // LDS-DMA kernels (hupr_k_wgrad_halo_glds / _m16), two gradients of ONE x in one launch (the two convolutions of a residual block read the same map):
    // output channels [0, co_split) take dy, [co_split, Co) take dy2 (same shape and stride); co_split == 0: one tensor.
    const void* dy2;
    int co_split;
};

constexpr int kRowB = 128;                   // bytes per LDS row: 64 bf16 channels

// one transpose-read pair (4 rows each) -> 8 consecutive K values (voxels) of this lane's column
__device__ __forceinline__ bf16x8 tr_pair(const char* base, int off0, int off1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + off1));
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo;
    u.s.b = hi;
    return u.v;
}

// LDS images are row-major [voxel][64 channels] with the 64-byte half of a row XOR-swapped by bit 1 of the row index:
// a 16-lane transpose-read group touches four consecutive rows x one 32-byte segment, and rows r, r + 2 (128-byte
// pitch, 64 banks x 4 B) would otherwise land on the same banks (measured: 44 % of the LDS cycles were conflicts).
__device__ __forceinline__ int swz_col(int row, int col_bytes) { return col_bytes ^ (((row >> 1) & 1) << 6); }

template <bool ABF, bool IS3D>
__global__ __launch_bounds__(256, 2) void hupr_k_wgrad_halo_bf16(WgradHaloArgs p) {   // 2 workgroups per CU: <= 112 VGPRs + 144 accumulators
    constexpr int TD = IS3D ? 2 : 1, TW = IS3D ? 8 : 16, LOG2TW = IS3D ? 3 : 4;
    constexpr int HH = 10, HW = TW + 2;
    constexpr int NVOX = TD * HH * HW;               // x halo of ONE depth-tap plane: 200 (3-D) / 180 (2-D) voxels
    constexpr int ITEMS_X = NVOX * 8, ITEMS = ITEMS_X + 128 * 8;
    __shared__ __attribute__((aligned(16))) char Xh[NVOX * kRowB];
    __shared__ __attribute__((aligned(16))) char DYs[128 * kRowB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;        // 32-row (co) / 32-col (ci) quadrant of the 64x64 tile
    const int pd = p.kd >> 1;
    const int T = p.kd * 9;

    const int group = blockIdx.x;
    const int td = blockIdx.y;                        // depth tap plane handled by this workgroup
    const int cot = blockIdx.z / p.n_ci_tiles, cit = blockIdx.z % p.n_ci_tiles;
    const int co0 = cot * 64, ci0 = cit * 64;

    // transpose-read supplier role of this lane: g = 16-lane group, s = index inside it
    const int g = lane >> 4, s = lane & 15;
    const int kh_ = g >> 1;                           // K half served by this lane (== lane >> 5)
    const int colb = (16 * (g & 1) + 4 * (s & 3)) * 2;   // byte offset of the 4-column segment inside a 32-col half
    const int rsub = s >> 2;                          // row (0..3) inside the 4-row group
    // Per-lane LDS byte offsets; everything that depends on (K-step, tap) is a compile-time immediate on top of them.
    //   dy rows: tile voxel v = 16 ks + c_t,  c_t = 8 kh + 4 t + rsub
    //   x rows : halo voxel R = lane_row(c_t) + kx + [ks rows + ky HW], whose swizzle bit is f(lane_row + kx) ^ par
    int dyb[2], xb[3][2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int c = 8 * kh_ + 4 * t + rsub;
        dyb[t] = c * kRowB + swz_col(c, wm * 64 + colb);
        const int lrow = IS3D ? ((c >> 3) * HW + (c & 7)) : c;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int par = 0; par < 2; ++par)
                xb[kx][par][t] = (lrow + kx) * kRowB + ((wn * 64 + colb) ^ (((((lrow + kx) >> 1) & 1) ^ par) << 6));
    }

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    constexpr int NI = (ITEMS + 255) / 256;          // 11 (3-D) / 10 (2-D) 8-channel items per thread per tile
    constexpr int NB = (NI + 1) / 2;                // two batches: a single 11-item batch costs 135 VGPRs + 144 accumulators (one wave/SIMD)
    for (int st = group; st < p.n_spatial; st += p.groups) {
        int q = st;
        const int twi = q % p.nw; q /= p.nw;
        const int thi = q % p.nh; q /= p.nh;
        const int tdi = q % p.nd;
        const int b = q / p.nd;
        const int d0 = tdi * TD, h0 = thi * 8, w0 = twi * TW;

        __syncthreads();                               // previous tile's fragments are consumed
        // ---- stage the x halo (this td plane) and the dy tile: branch-free clamped loads, zero-select, swizzled rows ----
#pragma unroll
        for (int ub = 0; ub < NI; ub += NB) {
            float4 va[ABF ? 1 : NB], vc[ABF ? 1 : NB];
            u32x4 vb[ABF ? NB : 1];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int it = tid + (ub + u) * 256;
                int off = 0;                             // element offsets fit 31 bits (checked by the launcher)
                bool ok = false, is_x = true;
                if (it < ITEMS_X) {
                    const int vox = it >> 3, c8 = it & 7;
                    const int hx = vox % HW;
                    const int t2 = vox / HW;
                    const int hy = t2 % HH, hz = t2 / HH;
                    const int d = d0 + hz + td - pd, h = h0 + hy - 1, w = w0 + hx - 1;
                    ok = (unsigned)d < (unsigned)p.D && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W &&
                         ci0 + c8 * 8 < p.Ci;       // Cin = 32 layers: the upper half of the 64-wide tile is zero
                    off = (((b * p.D + d) * p.H + h) * p.W + w) * p.in_ld + ci0 + c8 * 8;
                } else if (it < ITEMS) {
                    const int j = it - ITEMS_X;
                    const int v = j >> 3, c8 = j & 7;
                    const int wx = v & (TW - 1), hy = (v >> LOG2TW) & 7, dz = v >> (LOG2TW + 3);
                    is_x = false;
                    ok = co0 + c8 * 8 < p.Co;
                    off = (((b * p.D + d0 + dz) * p.H + h0 + hy) * p.W + w0 + wx) * p.dy_ld + co0 + c8 * 8;
                }
                if (!ok) off = 0;                      // any valid address; the value is replaced by zeros below
                if constexpr (ABF) {
                    const u32x4 ld = *reinterpret_cast<const u32x4*>(static_cast<const __bf16*>(is_x ? p.x : p.dy) + off);
                    vb[u] = ok ? ld : (u32x4){0u, 0u, 0u, 0u};
                } else {
                    const float* src = static_cast<const float*>(is_x ? p.x : p.dy) + off;
                    const float4 l0 = *reinterpret_cast<const float4*>(src), l1 = *reinterpret_cast<const float4*>(src + 4);
                    va[u] = ok ? l0 : make_float4(0.f, 0.f, 0.f, 0.f);
                    vc[u] = ok ? l1 : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int it = tid + (ub + u) * 256;
                if (it < ITEMS) {
                    const int row = (it < ITEMS_X) ? (it >> 3) : ((it - ITEMS_X) >> 3);
                    char* dstp = ((it < ITEMS_X) ? Xh : DYs) + row * kRowB + swz_col(row, (it & 7) * 16);
                    if constexpr (ABF) {
                        *reinterpret_cast<u32x4*>(dstp) = vb[u];
                    } else {
                        bf16x8 v;
                        v[0] = (__bf16)va[u].x; v[1] = (__bf16)va[u].y; v[2] = (__bf16)va[u].z; v[3] = (__bf16)va[u].w;
                        v[4] = (__bf16)vc[u].x; v[5] = (__bf16)vc[u].y; v[6] = (__bf16)vc[u].z; v[7] = (__bf16)vc[u].w;
                        *reinterpret_cast<bf16x8*>(dstp) = v;
                    }
                }
            }
        }
        __syncthreads();

        // ---- 8 K-steps of 16 voxels x 9 taps, as 24 groups of three taps (one ky row); the fragments of group j + 1
        // are read while group j multiplies; the dy fragment of a K-step is shared by its 9 taps --------------------
        bf16x8 a[2], xq[2][3];
#define HUPR_WG_LOAD(SET_, J_)                                                                                      \
        {                                                                                                           \
            constexpr int ks_ = (J_) / 3, ky_ = (J_) % 3;                                                           \
            constexpr int rows_ = (IS3D ? ((ks_ >> 2) * HH * HW + 2 * (ks_ & 3) * HW) : ks_ * HW) + ky_ * HW;       \
            constexpr int par_ = IS3D ? (ky_ & 1) : ((ks_ + ky_) & 1);                                              \
            _Pragma("unroll") for (int kx = 0; kx < 3; ++kx)                                                        \
                xq[SET_][kx] = tr_pair(Xh + rows_ * kRowB, xb[kx][par_][0], xb[kx][par_][1]);                       \
            if (ky_ == 0) a[ks_ & 1] = tr_pair(DYs + ks_ * 16 * kRowB, dyb[0], dyb[1]);                             \
        }
#define HUPR_WG_STEP(J_)                                                                                            \
        {                                                                                                           \
            if ((J_) + 1 < 24) { HUPR_WG_LOAD(((J_) + 1) & 1, ((J_) + 1 < 24 ? (J_) + 1 : 0)) }                      \
            __builtin_amdgcn_sched_barrier(0);                                                                      \
            _Pragma("unroll") for (int kx = 0; kx < 3; ++kx)                                                        \
                acc[((J_) % 3) * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[((J_) / 3) & 1], xq[(J_) & 1][kx], \
                                                                                   acc[((J_) % 3) * 3 + kx], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);                                                                      \
        }
        HUPR_WG_LOAD(0, 0)
        HUPR_WG_STEP(0) HUPR_WG_STEP(1) HUPR_WG_STEP(2) HUPR_WG_STEP(3) HUPR_WG_STEP(4) HUPR_WG_STEP(5)
        HUPR_WG_STEP(6) HUPR_WG_STEP(7) HUPR_WG_STEP(8) HUPR_WG_STEP(9) HUPR_WG_STEP(10) HUPR_WG_STEP(11)
        HUPR_WG_STEP(12) HUPR_WG_STEP(13) HUPR_WG_STEP(14) HUPR_WG_STEP(15) HUPR_WG_STEP(16) HUPR_WG_STEP(17)
        HUPR_WG_STEP(18) HUPR_WG_STEP(19) HUPR_WG_STEP(20) HUPR_WG_STEP(21) HUPR_WG_STEP(22) HUPR_WG_STEP(23)
#undef HUPR_WG_LOAD
#undef HUPR_WG_STEP
    }

    // ---- partial[group][co][tap][ci]; D layout: col = lane&31 (ci), row = (r&3)+8*(r>>2)+4*(lane>>5) (co) ------
    const int lr = lane & 31, lh = lane >> 5;
    const int ci = ci0 + wn * 32 + lr;
    float* part = p.part + (long)group * p.Co * T * p.Ci;
    if (ci < p.Ci) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co < p.Co) part[((long)co * T + td * 9 + tap) * p.Ci + ci] = acc[tap][r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16-activation variant with LDS-DMA staging.  The register-staged kernel above is VGPR-capped (144 accumulators +
// 2 waves/SIMD leave no room to prefetch a tile), so its fill is a synchronous round trip per 128 voxels.  Here:
//   * x halo and dy tile travel global -> LDS by buffer_load_dwordx4 ... lds (no VGPRs, out-of-range lanes deposit
//     zeros = the conv padding; probe: scripts/probes/glds_probe.hip) into the idle half of a double-buffered image
//     while the MFMAs run on the other half; one barrier per tile;
//   * the row swizzle is applied on the SOURCE side (LDS slot = base + lane * 16 is fixed): slot (row, c') is filled
//     with channel chunk c' ^ (4 * bit1(row));
//   * 512 threads: waves 0-3 multiply K-steps 0..3 of the tile, waves 4-7 K-steps 4..7, each half keeping its own
//     9 x (32x32) accumulators and writing its own partial tensor (partial index 2 * group + half).
// ---------------------------------------------------------------------------------------------------------------------
//   * CI32 (Ci <= 32, one ci quadrant): the wn = 1 waves would multiply zero columns; instead the wave's wn bit splits K
//     once more (K quarter = 2 kq + wn: two K-steps of the tile each) and the four partial accumulator sets are merged in
//     LDS at the end — half the MFMAs per tile for the 32-channel input of Encoder3D.layer1's first convolution.
template <bool IS3D, bool CI32>
__global__ __launch_bounds__(512) void hupr_k_wgrad_halo_glds(WgradHaloArgs p) {
    constexpr bool g_sched_burst_ = false;      // true: the round-1 order (all reads of a group, then its MFMAs)
    constexpr int TD = IS3D ? 2 : 1, TW = IS3D ? 8 : 16, LOG2TW = IS3D ? 3 : 4;
    constexpr int HH = 10, HW = TW + 2;
    constexpr int NVOX = TD * HH * HW;               // 200 (3-D) / 180 (2-D) halo voxels of one depth-tap plane
    constexpr int NVOXP = (NVOX + 7) / 8 * 8;        // x rows padded to whole 1 KiB DMA pieces (8 rows)
    constexpr int ITEMS_X = NVOXP * 8, ITEMS = ITEMS_X + 128 * 8;
    constexpr int IMG = (NVOXP + 128) * kRowB;       // one staged tile: x halo rows, then the dy rows
    // THREE DISTINCT LDS objects with static roles in a by-3 unrolled loop: hipcc's waitcnt pass then proves that the image
    // being read is not the target of a pending LDS-DMA and emits counted vmcnt(N) instead of vmcnt(0) in front of the
    // first ds_read after an issue (with one array and a rotating index every tile waited for the DMA it had just issued)
    __shared__ __attribute__((aligned(1024))) char bufA[IMG];
    __shared__ __attribute__((aligned(1024))) char bufB[IMG];
    __shared__ __attribute__((aligned(1024))) char bufC[IMG];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = wave >> 2;                         // K half of this wave
    const int wm = (wave >> 1) & 1, wbit = wave & 1;  // 32-row (co) quadrant; wbit: 32-col (ci) quadrant, or K quarter (CI32)
    const int wn = CI32 ? 0 : wbit;
    const int pd = p.kd >> 1;
    const int T = p.kd * 9;
    // XCD-aware 1-D grid (p.xcd_map): the kd depth-tap planes and the (co, ci) tile pairs of one spatial group re-read the
    // same x / dy tiles, so all of them are given workgroup ids that are congruent mod 8 (= one XCD, one L2): measured on
    // the layer-1 shape the HBM fetch drops from 3.1x to 1.1x of the algorithmic bytes.  id = c + 8 n, c = group % 8,
    // n = (group / 8) * members + member.  The launcher only chooses it when members * groups / 8 <= 32 workgroups per XCD
    // still fill >= 90 % of the CUs; otherwise the grid is (groups, kd, tile pairs) as before.
    int group, td, pair_;
    if (p.xcd_map) {
        const int members = p.kd * p.n_ci_tiles * p.n_co_tiles;
        const int slot = (int)blockIdx.x >> 3;
        group = (slot / members) * 8 + ((int)blockIdx.x & 7);
        const int member = slot % members;
        td = member % p.kd;
        pair_ = member / p.kd;
    } else {
        group = blockIdx.x;
        td = blockIdx.y;
        pair_ = blockIdx.z;
    }
    const int cot = pair_ / p.n_ci_tiles, cit = pair_ % p.n_ci_tiles;
    if (group >= p.groups) return;
    const int co0 = cot * 64, ci0 = cit * 64;
    // which gradient tensor this workgroup's 64 output channels come from (see WgradHaloArgs::co_split)
    const bool second = p.co_split != 0 && co0 >= p.co_split;
    const void* const dyp = second ? p.dy2 : p.dy;
    const int co0d = second ? co0 - p.co_split : co0, Cod = p.co_split ? p.co_split : p.Co;

    const int g = lane >> 4, s = lane & 15;
    const int kh_ = g >> 1;
    const int colb = (16 * (g & 1) + 4 * (s & 3)) * 2;
    const int rsub = s >> 2;
    int dyb[2], xb[3][2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int c = 8 * kh_ + 4 * t + rsub;
        dyb[t] = NVOXP * kRowB + c * kRowB + swz_col(c, wm * 64 + colb);
        const int lrow = IS3D ? ((c >> 3) * HW + (c & 7)) : c;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int par = 0; par < 2; ++par)
                xb[kx][par][t] = (lrow + kx) * kRowB + ((wn * 64 + colb) ^ (((((lrow + kx) >> 1) & 1) ^ par) << 6));
    }

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const long x_bytes = (long)p.Bn * p.D * p.H * p.W * p.in_ld * 2, dy_bytes = (long)p.Bn * p.D * p.H * p.W * p.dy_ld * 2;
    constexpr int kOOB = 0x7ffffff0;                 // beyond num_records: the DMA deposits zeros

    // LDS-DMA pieces of this wave: piece pc = 8 u + wave moves 64 items (8 rows x 128 B) to image offset pc KiB.  What
    // does not depend on the tile is computed once: the byte offset of the lane's 16 bytes relative to the tile origin
    // (rel) and which tile borders would put it outside the tensor (msk; bit 6 = always outside: row padding / channel
    // tail).  Per tile a piece then costs an add, a mask test and a select.
    constexpr int NP = (ITEMS / 64 + 7) / 8;          // 6 pieces for the first wave(s), 5 for the others (3-D)
    constexpr int NLAST = ITEMS / 64 - (NP - 1) * 8;  // waves that own a piece in the last round
    constexpr int PX = ITEMS_X / 64;                  // pieces [0, PX) are x halo rows, the rest dy rows
    int rel[NP], msk[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int it = tid + u * 512;
        rel[u] = 0;
        msk[u] = 64;
        if (it < ITEMS_X) {
            const int vox = it >> 3, c8 = (it & 7) ^ (((vox >> 1) & 1) << 2);
            const int hx = vox % HW;
            const int t2 = vox / HW;
            const int hy = t2 % HH, hz = t2 / HH;
            const int dzr = hz + td - pd, hyr = hy - 1, hxr = hx - 1;
            rel[u] = (((dzr * p.H + hyr) * p.W + hxr) * p.in_ld + ci0 + c8 * 8) * 2;
            msk[u] = (dzr < 0 ? 1 : 0) | (dzr >= TD ? 2 : 0) | (hyr < 0 ? 4 : 0) | (hyr >= 8 ? 8 : 0) | (hxr < 0 ? 16 : 0) |
                     (hxr >= TW ? 32 : 0) | ((vox >= NVOX || ci0 + c8 * 8 >= p.Ci) ? 64 : 0);
        } else if (it < ITEMS) {
            const int j = it - ITEMS_X;
            const int v = j >> 3, c8 = (j & 7) ^ (((v >> 1) & 1) << 2);
            const int wx = v & (TW - 1), hy = (v >> LOG2TW) & 7, dz = v >> (LOG2TW + 3);
            rel[u] = (((dz * p.H + hy) * p.W + wx) * p.dy_ld + co0d + c8 * 8) * 2;
            msk[u] = (co0d + c8 * 8 >= Cod) ? 64 : 0;
        }
    }
    // The fill travels by LDS-DMA issued from inline asm (M0 = LDS address of the wave's 1 KiB piece, saved / restored around it):
    // hipcc then neither counts it nor guards LDS reads with vmcnt waits of its own — the tile protocol below does its own counted
    // waits (rounds 1-4 used the builtin and three LDS objects with static roles so that hipcc's bookkeeping came out right; that
    // form could not carry a fill that is spread over the MFMA groups).  The fill is UNCONDITIONAL (past the last tile it
    // re-fetches the last one), so every wave always has the same number of pieces in flight.
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const u32x4 rxs = {(unsigned)(unsigned long)p.x, (unsigned)((unsigned long)p.x >> 32) & 0xffffu, (unsigned)x_bytes, 0x00020000u};
    const u32x4 rdys = {(unsigned)(unsigned long)dyp, (unsigned)((unsigned long)dyp >> 32) & 0xffffu, (unsigned)dy_bytes, 0x00020000u};
    // tile coordinates of the fill advance by carries (p.groups in mixed radix), not by divisions
    const int last_tile = group + ((p.n_spatial - 1 - group) / p.groups) * p.groups;      // last tile of this workgroup
    int gw_, gh_, gd_, gb_;
    {
        int t_ = p.groups;
        gw_ = t_ % p.nw; t_ /= p.nw;
        gh_ = t_ % p.nh; t_ /= p.nh;
        gd_ = t_ % p.nd;
        gb_ = t_ / p.nd;
    }
    int f_t = group, f_twi, f_thi, f_tdi, f_b;
    {
        int q_ = group;
        f_twi = q_ % p.nw; q_ /= p.nw;
        f_thi = q_ % p.nh; q_ /= p.nh;
        f_tdi = q_ % p.nd;
        f_b = q_ / p.nd;
    }
    int fl_bx = 0, fl_bdy = 0, fl_flags = 0;
    // coordinates of the tile the NEXT fill carries (then advance to the one after)
#define HUPR_WG_FILL_OPEN()                                                                                         \
    {                                                                                                               \
        const int d0_ = f_tdi * TD, h0_ = f_thi * 8, w0_ = f_twi * TW;                                              \
        const int org_ = ((f_b * p.D + d0_) * p.H + h0_) * p.W + w0_;                                               \
        fl_bx = org_ * p.in_ld * 2;                                                                                 \
        fl_bdy = org_ * p.dy_ld * 2;                                                                                \
        fl_flags = 64 | (d0_ == 0 ? 1 : 0) | (d0_ + TD == p.D ? 2 : 0) | (h0_ == 0 ? 4 : 0) |                       \
                   (h0_ + 8 == p.H ? 8 : 0) | (w0_ == 0 ? 16 : 0) | (w0_ + TW == p.W ? 32 : 0);                     \
        if (f_t + p.groups <= last_tile) {                                                                          \
            f_t += p.groups;                                                                                        \
            f_twi += gw_;                                                                                           \
            int c_ = f_twi >= p.nw ? 1 : 0;                                                                         \
            f_twi -= c_ ? p.nw : 0;                                                                                 \
            f_thi += gh_ + c_;                                                                                      \
            c_ = f_thi >= p.nh ? 1 : 0;                                                                             \
            f_thi -= c_ ? p.nh : 0;                                                                                 \
            f_tdi += gd_ + c_;                                                                                      \
            c_ = f_tdi >= p.nd ? 1 : 0;                                                                             \
            f_tdi -= c_ ? p.nd : 0;                                                                                 \
            f_b += gb_ + c_;                                                                                        \
        }                                                                                                           \
    }
    // piece U_ (compile-time) of the open fill -> image BUF_
#define HUPR_WG_PIECE(U_, BUF_)                                                                                     \
    if ((U_) < NP - 1 || wave_u < NLAST) {                                                                          \
        const int pc_ = (U_) * 8 + wave_u;                       /* wave-uniform */                                 \
        const bool isx_ = pc_ < PX;                                                                                 \
        const int voff_ = (msk[U_] & fl_flags) ? kOOB : (isx_ ? fl_bx : fl_bdy) + rel[U_];                          \
        const unsigned dst_ = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)(BUF_) + pc_ * 1024; \
        unsigned keep_;                                                                                             \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\t" \
                     "s_mov_b32 m0, %0"                                                                             \
                     : "=&s"(keep_)                                                                                 \
                     : "s"(dst_), "v"(voff_), "s"(isx_ ? rxs : rdys)                                                \
                     : "memory");                                                                                   \
    }

    bf16x8 a[2], xq[2][3];
    // K-steps KS0_ .. KS0_+3 of the staged tile IMG_ as 12 groups of three taps, fragments one group ahead
#define HUPR_WG2_LOAD(IMG_, SET_, KS0_, J_)                                                                         \
    {                                                                                                               \
        constexpr int ks_ = (KS0_) + (J_) / 3, ky_ = (J_) % 3;                                                      \
        constexpr int rows_ = (IS3D ? ((ks_ >> 2) * HH * HW + 2 * (ks_ & 3) * HW) : ks_ * HW) + ky_ * HW;           \
        constexpr int par_ = IS3D ? (ky_ & 1) : ((ks_ + ky_) & 1);                                                  \
        _Pragma("unroll") for (int kx = 0; kx < 3; ++kx)                                                            \
            xq[SET_][kx] = tr_pair((IMG_) + rows_ * kRowB, xb[kx][par_][0], xb[kx][par_][1]);                       \
        if (ky_ == 0) a[ks_ & 1] = tr_pair((IMG_) + ks_ * 16 * kRowB, dyb[0], dyb[1]);                              \
    }
    // group J_ of NJ_: its three MFMAs; under them the next group's fragments — of this tile (J_ + 1 < NJ_) or group 0 of the
    // tile in NXT_ — and, in the first groups, one piece each of the open fill -> FREE_
#define HUPR_WG2_STEP(IMG_, NXT_, FREE_, KS0_, J_, NJ_)                                                             \
    {                                                                                                               \
        if ((J_) + 1 < (NJ_)) { HUPR_WG2_LOAD(IMG_, ((J_) + 1) & 1, KS0_, ((J_) + 1 < (NJ_) ? (J_) + 1 : 0)) }      \
        else { HUPR_WG2_LOAD(NXT_, 0, KS0_, 0) }                                                                    \
        if ((J_) + 1 < (NJ_)) {                                                                                     \
            _Pragma("unroll") for (int u_ = 0; u_ < NP; ++u_)                                                       \
                if (u_ % ((NJ_) - 1) == (J_)) { HUPR_WG_PIECE(u_, FREE_) }                                          \
        }                                                                                                           \
        _Pragma("unroll") for (int kx = 0; kx < 3; ++kx)                                                            \
            acc[((J_) % 3) * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[((KS0_) + (J_) / 3) & 1], xq[(J_) & 1][kx], \
                                                                               acc[((J_) % 3) * 3 + kx], 0, 0, 0);  \
        /* the next group's fragment reads spread between this group's MFMAs */                                     \
        _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) {                                                          \
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                      \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                      \
        }                                                                                                           \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
    }
    // all groups but the last of a half tile (6 groups) / tile (12 groups)
#define HUPR_WG2_HEAD6(IMG_, NXT_, FREE_, KS0_)                                                                     \
    HUPR_WG2_STEP(IMG_, NXT_, FREE_, KS0_, 0, 6) HUPR_WG2_STEP(IMG_, NXT_, FREE_, KS0_, 1, 6) HUPR_WG2_STEP(IMG_, NXT_, FREE_, KS0_, 2, 6) \
    HUPR_WG2_STEP(IMG_, NXT_, FREE_, KS0_, 3, 6) HUPR_WG2_STEP(IMG_, NXT_, FREE_, KS0_, 4, 6)
#define HUPR_WG2_HEAD12(IMG_, NXT_, FREE_, KS0_)                                                                    \
    HUPR_WG2_STEP(IMG_, NXT_, FREE_, KS0_, 0, 12) HUPR_WG2_STEP(IMG_, NXT_, FREE_, KS0_, 1, 12) HUPR_WG2_STEP(IMG_, NXT_, FREE_, KS0_, 2, 12) \
    HUPR_WG2_STEP(IMG_, NXT_, FREE_, KS0_, 3, 12) HUPR_WG2_STEP(IMG_, NXT_, FREE_, KS0_, 4, 12) HUPR_WG2_STEP(IMG_, NXT_, FREE_, KS0_, 5, 12) \
    HUPR_WG2_STEP(IMG_, NXT_, FREE_, KS0_, 6, 12) HUPR_WG2_STEP(IMG_, NXT_, FREE_, KS0_, 7, 12) HUPR_WG2_STEP(IMG_, NXT_, FREE_, KS0_, 8, 12) \
    HUPR_WG2_STEP(IMG_, NXT_, FREE_, KS0_, 9, 12) HUPR_WG2_STEP(IMG_, NXT_, FREE_, KS0_, 10, 12)

    // Tile protocol (round 5; the SQ counters of the old one — barrier, whole fill, first fragments, then the MFMAs — showed the
    // matrix pipe busy 53 % of the cycles: profiles/r04_wgrad_sq_pmc.txt).  Three images in a ring: CUR (tile st), NXT (tile
    // st + 1, landing or landed), FREE (tile st - 1's, released by the previous barrier).  A wave multiplies all groups of tile st
    // but the last, issuing one piece of the fill of tile st + 2 -> FREE under each of the first groups; then it waits until its
    // own pieces of tile st + 1 have landed (all but the youngest NP or NP - 1 operations: those of tile st + 2) and its reads of
    // CUR have returned, and takes the tile's ONE barrier holding the last group's fragments: it leaves the barrier with three
    // MFMAs ready and reads the first fragments of tile st + 1 under them.  The loop starts two (virtual) tiles early with the
    // multiply switched off.
#define HUPR_WG_ITER(CUR_, NXT_, FREE_)                                                                             \
    {                                                                                                               \
        if (st >= p.n_spatial) break;                                                                               \
        HUPR_WG_FILL_OPEN()                                                                                         \
        if (st >= 0) {                                                                                              \
            if (CI32) {                                                                                             \
                if (kq == 0) { if (wbit == 0) { HUPR_WG2_HEAD6(CUR_, NXT_, FREE_, 0) } else { HUPR_WG2_HEAD6(CUR_, NXT_, FREE_, 2) } } \
                else { if (wbit == 0) { HUPR_WG2_HEAD6(CUR_, NXT_, FREE_, 4) } else { HUPR_WG2_HEAD6(CUR_, NXT_, FREE_, 6) } } \
            } else {                                                                                                \
                if (kq == 0) { HUPR_WG2_HEAD12(CUR_, NXT_, FREE_, 0) } else { HUPR_WG2_HEAD12(CUR_, NXT_, FREE_, 4) } \
            }                                                                                                       \
        } else {                                                                                                    \
            _Pragma("unroll") for (int u_ = 0; u_ < NP; ++u_) { HUPR_WG_PIECE(u_, FREE_) }                          \
        }                                                                                                           \
        if (wave_u < NLAST) __builtin_amdgcn_s_waitcnt(0x0070 | NP);                                                \
        else __builtin_amdgcn_s_waitcnt(0x0070 | (NP - 1));                                                         \
        __builtin_amdgcn_s_barrier();                                                                               \
        asm volatile("" ::: "memory");                                                                              \
        if (st >= 0) {                                                                                              \
            if (CI32) {                                                                                             \
                if (kq == 0) { if (wbit == 0) { HUPR_WG2_STEP(CUR_, NXT_, FREE_, 0, 5, 6) } else { HUPR_WG2_STEP(CUR_, NXT_, FREE_, 2, 5, 6) } } \
                else { if (wbit == 0) { HUPR_WG2_STEP(CUR_, NXT_, FREE_, 4, 5, 6) } else { HUPR_WG2_STEP(CUR_, NXT_, FREE_, 6, 5, 6) } } \
            } else {                                                                                                \
                if (kq == 0) { HUPR_WG2_STEP(CUR_, NXT_, FREE_, 0, 11, 12) } else { HUPR_WG2_STEP(CUR_, NXT_, FREE_, 4, 11, 12) } \
            }                                                                                                       \
        } else if (st + p.groups >= 0) {                        /* the first real tile's first fragments */         \
            if (CI32) {                                                                                             \
                if (kq == 0) { if (wbit == 0) { HUPR_WG2_LOAD(NXT_, 0, 0, 0) } else { HUPR_WG2_LOAD(NXT_, 0, 2, 0) } } \
                else { if (wbit == 0) { HUPR_WG2_LOAD(NXT_, 0, 4, 0) } else { HUPR_WG2_LOAD(NXT_, 0, 6, 0) } }      \
            } else {                                                                                                \
                if (kq == 0) { HUPR_WG2_LOAD(NXT_, 0, 0, 0) } else { HUPR_WG2_LOAD(NXT_, 0, 4, 0) }                 \
            }                                                                                                       \
        }                                                                                                           \
        st += p.groups;                                                                                             \
    }
    static_assert(NP < 16 && NP <= 11, "piece count exceeds the counted wait / the groups of a tile");
    if (group >= p.n_spatial) return;
    int st = group - 2 * p.groups;
    for (;;) {
        HUPR_WG_ITER(bufB, bufC, bufA)
        HUPR_WG_ITER(bufC, bufA, bufB)
        HUPR_WG_ITER(bufA, bufB, bufC)
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);              // the two re-fetches past the last tile must land before the LDS is released
#undef HUPR_WG_ITER
#undef HUPR_WG_FILL_OPEN
#undef HUPR_WG_PIECE
#undef HUPR_WG2_LOAD
#undef HUPR_WG2_STEP
#undef HUPR_WG2_HEAD6
#undef HUPR_WG2_HEAD12

    // Merge the two K halves through the (now dead) images: the kq = 1 waves park their accumulators, six taps and then
    // three, the kq = 0 waves add them — one partial tensor per workgroup instead of two halves what the workgroups
    // write at the very end of the kernel and what the split-K reduction reads back.
    {
        const int t256 = tid & 255;
        float* const red[3] = {reinterpret_cast<float*>(bufA), reinterpret_cast<float*>(bufB), reinterpret_cast<float*>(bufC)};
        __syncthreads();
        if (CI32) {                                  // K quarters first: the wbit = 1 waves park, their wbit = 0 partners add
            const int q256 = (wave >> 1) * 64 + lane;
#pragma unroll
            for (int round = 0; round < 2; ++round) {
                if (round) __syncthreads();
                if (wbit == 1) {
#pragma unroll
                    for (int tap = round * 6; tap < (round ? 9 : 6); ++tap)
#pragma unroll
                        for (int r = 0; r < 16; ++r) red[((tap - round * 6) >> 1)][(((tap & 1) * 16 + r) << 8) + q256] = acc[tap][r];
                }
                __syncthreads();
                if (wbit == 0) {
#pragma unroll
                    for (int tap = round * 6; tap < (round ? 9 : 6); ++tap)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[tap][r] += red[((tap - round * 6) >> 1)][(((tap & 1) * 16 + r) << 8) + q256];
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            if (round) __syncthreads();
            if (kq == 1) {
#pragma unroll
                for (int tap = round * 6; tap < (round ? 9 : 6); ++tap)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((tap - round * 6) >> 1)][(((tap & 1) * 16 + r) << 8) + t256] = acc[tap][r];
            }
            __syncthreads();
            if (kq == 0) {
#pragma unroll
                for (int tap = round * 6; tap < (round ? 9 : 6); ++tap)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tap][r] += red[((tap - round * 6) >> 1)][(((tap & 1) * 16 + r) << 8) + t256];
            }
        }
    }
    if (kq != 0 || (CI32 && wbit != 0)) return;
    const int lr = lane & 31, lh = lane >> 5;
    const int ci = ci0 + wn * 32 + lr;
    float* part = p.part + (long)group * p.Co * T * p.Ci;
    if (ci < p.Ci) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co < p.Co) part[((long)co * T + td * 9 + tap) * p.Ci + ci] = acc[tap][r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The LDS-DMA weight-gradient kernel on v_mfma_f32_16x16x32_bf16 (round 5).  At the chip's power limit the
// 16 x 16 x 32 form does 15-17 % more work per joule than 32 x 32 x 16 (scripts/probes/mfma_shapes_probe.hip); the forward /
// input-gradient convolution gained 6-7 % from the same move in round 4.  Same tile (128 voxels: 2 x 8 x 8 or 1 x 8 x 16; one
// depth-tap plane; 64 x 64 weights), same three-image LDS-DMA ring and tile protocol, same K-half merge and partial-sum layout as
// hupr_k_wgrad_halo_glds; what changes is who owns what:
//   wave = K half kq (two 32-voxel K-steps of the tile) x 16-wide ci block cb; it multiplies ALL FOUR 16-wide co blocks: per
//   (K-step, ky) three x fragments (kx) and — once per K-step — four dy fragments feed twelve MFMAs (the 32 x 32 wave tile of the old
//   kernel read 40 transpose fragments per 32 voxels, this one 26);
//   lane ownership inside a K-step and the two-bit row swizzle: see the comment at the address tables below.
// ---------------------------------------------------------------------------------------------------------------------
typedef float f32x4w __attribute__((ext_vector_type(4)));

// CI32 (round 6; Ci <= 32: the 32-channel input of Encoder3D.layer1's first convolution, before on hupr_k_wgrad_halo_glds<.., true> at
// 806 TF/s): only two of the four 16-wide ci blocks exist, so a wave is a K QUARTER (one 32-voxel K-step of the tile: three groups
// of twelve MFMAs) x ci block 0 / 1, and the four quarters are merged through the dead images at the end (fixed order).
template <bool IS3D, bool CI32 = false>
__global__ __launch_bounds__(512) void hupr_k_wgrad_halo_m16(WgradHaloArgs p) {
    constexpr int TD = IS3D ? 2 : 1, TW = IS3D ? 8 : 16, LOG2TW = IS3D ? 3 : 4;
    constexpr int HH = 10, HW = TW + 2;
    constexpr int NVOX = TD * HH * HW;               // 200 (3-D) / 180 (2-D) halo voxels of one depth-tap plane
    constexpr int NVOXP = (NVOX + 7) / 8 * 8;        // x rows padded to whole 1 KiB DMA pieces (8 rows)
    constexpr int ITEMS_X = NVOXP * 8, ITEMS = ITEMS_X + 128 * 8;
    constexpr int IMG = (NVOXP + 128) * kRowB;       // one staged tile: x halo rows, then the dy rows
    // THREE DISTINCT LDS objects with static roles in a by-3 unrolled loop: hipcc's waitcnt pass then proves that the image
    // being read is not the target of a pending LDS-DMA and emits counted vmcnt(N) instead of vmcnt(0) in front of the
    // first ds_read after an issue (with one array and a rotating index every tile waited for the DMA it had just issued)
    __shared__ __attribute__((aligned(1024))) char bufA[IMG];
    __shared__ __attribute__((aligned(1024))) char bufB[IMG];
    __shared__ __attribute__((aligned(1024))) char bufC[IMG];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = CI32 ? wave >> 1 : wave >> 2;      // K half of this wave: K-steps 2 kq, 2 kq + 1 of the tile (32 voxels each); CI32: K quarter = K-step kq
    const int cb = CI32 ? wave & 1 : wave & 3;        // its 16-wide ci block of the 64-wide tile (all four 16-wide co blocks)
    const int pd = p.kd >> 1;
    const int T = p.kd * 9;
    // XCD-aware 1-D grid (p.xcd_map): the kd depth-tap planes and the (co, ci) tile pairs of one spatial group re-read the
    // same x / dy tiles, so all of them are given workgroup ids that are congruent mod 8 (= one XCD, one L2): measured on
    // the layer-1 shape the HBM fetch drops from 3.1x to 1.1x of the algorithmic bytes.  id = c + 8 n, c = group % 8,
    // n = (group / 8) * members + member.  The launcher only chooses it when members * groups / 8 <= 32 workgroups per XCD
    // still fill >= 90 % of the CUs; otherwise the grid is (groups, kd, tile pairs) as before.
    int group, td, pair_;
    if (p.xcd_map) {
        const int members = p.kd * p.n_ci_tiles * p.n_co_tiles;
        const int slot = (int)blockIdx.x >> 3;
        group = (slot / members) * 8 + ((int)blockIdx.x & 7);
        const int member = slot % members;
        td = member % p.kd;
        pair_ = member / p.kd;
    } else {
        group = blockIdx.x;
        td = blockIdx.y;
        pair_ = blockIdx.z;
    }
    const int cot = pair_ / p.n_ci_tiles, cit = pair_ % p.n_ci_tiles;
    if (group >= p.groups) return;
    const int co0 = cot * 64, ci0 = cit * 64;
    // which gradient tensor this workgroup's 64 output channels come from (see WgradHaloArgs::co_split)
    const bool second = p.co_split != 0 && co0 >= p.co_split;
    const void* const dyp = second ? p.dy2 : p.dy;
    const int co0d = second ? co0 - p.co_split : co0, Cod = p.co_split ? p.co_split : p.Co;

    // Lane (s = lane & 15, kq4 = lane >> 4) of a 16 x 16 x 32 operand owns column s (a co / ci of its block) and reduction elements
    // 8 kq4 .. 8 kq4 + 7 = two transpose reads (t = 0, 1) of four image rows each; as a SUPPLIER it addresses the 8-byte segment
    // 4 (s & 3) of row j = s >> 2 of its group's four.  Reduction element (kq4, t, j) is the voxel in row 2 (kq4 >> 1) + t, column
    // 4 (kq4 & 1) + j of the K-step's 4-row x 8-column block, so that the two 16-lane groups of a 32-lane read touch EIGHT CONSECUTIVE
    // image rows x one 32-byte segment; with the 32-byte granules of a row XOR-ed by (row >> 1) & 3 (swz2, source side of the fill)
    // those are all 64 banks exactly once, at every tap offset (scripts/lds_bank_model.py: 0 conflict cycles; the 32 x 32 x 16
    // kernel's ownership or its one-bit swizzle on this instruction shape: 100 % extra).
    const int s = lane & 15, kq4 = lane >> 4;
    const int colb = 8 * (s & 3);                      // byte offset of the supplier's 4-column segment inside a 16-column block
    int dyb[2][4], xa[8][2];                           // dy: [t][co block]; x halo: [row offset mod 8][t]
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int vr = 2 * (kq4 >> 1) + t, vc = 4 * (kq4 & 1) + (s >> 2);
        const int r_dy = vr * TW + vc;                 // (K-step offsets of the dy tile are multiples of 8 rows: the key is the lane's)
        // CI32: the wave's K-step kq (depth slice kq >> 1, row half kq & 1) is part of its tables — one code path for the four quarters
        // (a branch per quarter made hipcc keep a second set of accumulators)
        static_assert(!CI32 || IS3D, "K quarters: 3-D tiles only");
        const int q_dy = CI32 ? 32 * kq : 0, q_x = CI32 ? (kq >> 1) * HH * HW + 4 * (kq & 1) * HW : 0;
#pragma unroll
        for (int cob = 0; cob < 4; ++cob)
            dyb[t][cob] = (NVOXP + r_dy + q_dy) * kRowB + ((32 * cob + colb) ^ (((r_dy >> 1) & 3) << 5));
        const int r_x = vr * HW + vc + q_x;
#pragma unroll
        for (int m = 0; m < 8; ++m)
            xa[m][t] = r_x * kRowB + ((32 * cb + colb) ^ ((((r_x + m) >> 1) & 3) << 5));
    }

    f32x4w acc[9][4];                                  // [tap][co block]: rows 4 kq4 + i of the block, column s
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[t][c] = (f32x4w){0.f, 0.f, 0.f, 0.f};

    const long x_bytes = (long)p.Bn * p.D * p.H * p.W * p.in_ld * 2, dy_bytes = (long)p.Bn * p.D * p.H * p.W * p.dy_ld * 2;
    constexpr int kOOB = 0x7ffffff0;                 // beyond num_records: the DMA deposits zeros

    // LDS-DMA pieces of this wave: piece pc = 8 u + wave moves 64 items (8 rows x 128 B) to image offset pc KiB.  What
    // does not depend on the tile is computed once: the byte offset of the lane's 16 bytes relative to the tile origin
    // (rel) and which tile borders would put it outside the tensor (msk; bit 6 = always outside: row padding / channel
    // tail).  Per tile a piece then costs an add, a mask test and a select.
    constexpr int NP = (ITEMS / 64 + 7) / 8;          // 6 pieces for the first wave(s), 5 for the others (3-D)
    constexpr int NLAST = ITEMS / 64 - (NP - 1) * 8;  // waves that own a piece in the last round
    constexpr int PX = ITEMS_X / 64;                  // pieces [0, PX) are x halo rows, the rest dy rows
    int rel[NP], msk[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int it = tid + u * 512;
        rel[u] = 0;
        msk[u] = 64;
        if (it < ITEMS_X) {
            const int vox = it >> 3, c8 = (it & 7) ^ (((vox >> 1) & 3) << 1);      // 32-byte granule (c8 >> 1) ^ ((row >> 1) & 3)
            const int hx = vox % HW;
            const int t2 = vox / HW;
            const int hy = t2 % HH, hz = t2 / HH;
            const int dzr = hz + td - pd, hyr = hy - 1, hxr = hx - 1;
            rel[u] = (((dzr * p.H + hyr) * p.W + hxr) * p.in_ld + ci0 + c8 * 8) * 2;
            msk[u] = (dzr < 0 ? 1 : 0) | (dzr >= TD ? 2 : 0) | (hyr < 0 ? 4 : 0) | (hyr >= 8 ? 8 : 0) | (hxr < 0 ? 16 : 0) |
                     (hxr >= TW ? 32 : 0) | ((vox >= NVOX || ci0 + c8 * 8 >= p.Ci) ? 64 : 0);
        } else if (it < ITEMS) {
            const int j = it - ITEMS_X;
            const int v = j >> 3, c8 = (j & 7) ^ (((v >> 1) & 3) << 1);
            const int wx = v & (TW - 1), hy = (v >> LOG2TW) & 7, dz = v >> (LOG2TW + 3);
            rel[u] = (((dz * p.H + hy) * p.W + wx) * p.dy_ld + co0d + c8 * 8) * 2;
            msk[u] = (co0d + c8 * 8 >= Cod) ? 64 : 0;
        }
    }
    // The fill travels by LDS-DMA issued from inline asm (M0 = LDS address of the wave's 1 KiB piece, saved / restored around it):
    // hipcc then neither counts it nor guards LDS reads with vmcnt waits of its own — the tile protocol below does its own counted
    // waits (rounds 1-4 used the builtin and three LDS objects with static roles so that hipcc's bookkeeping came out right; that
    // form could not carry a fill that is spread over the MFMA groups).  The fill is UNCONDITIONAL (past the last tile it
    // re-fetches the last one), so every wave always has the same number of pieces in flight.
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const u32x4 rxs = {(unsigned)(unsigned long)p.x, (unsigned)((unsigned long)p.x >> 32) & 0xffffu, (unsigned)x_bytes, 0x00020000u};
    const u32x4 rdys = {(unsigned)(unsigned long)dyp, (unsigned)((unsigned long)dyp >> 32) & 0xffffu, (unsigned)dy_bytes, 0x00020000u};
    // tile coordinates of the fill advance by carries (p.groups in mixed radix), not by divisions
    const int last_tile = group + ((p.n_spatial - 1 - group) / p.groups) * p.groups;      // last tile of this workgroup
    int gw_, gh_, gd_, gb_;
    {
        int t_ = p.groups;
        gw_ = t_ % p.nw; t_ /= p.nw;
        gh_ = t_ % p.nh; t_ /= p.nh;
        gd_ = t_ % p.nd;
        gb_ = t_ / p.nd;
    }
    int f_t = group, f_twi, f_thi, f_tdi, f_b;
    {
        int q_ = group;
        f_twi = q_ % p.nw; q_ /= p.nw;
        f_thi = q_ % p.nh; q_ /= p.nh;
        f_tdi = q_ % p.nd;
        f_b = q_ / p.nd;
    }
    int fl_bx = 0, fl_bdy = 0, fl_flags = 0;
    // coordinates of the tile the NEXT fill carries (then advance to the one after)
#define HUPR_WG_FILL_OPEN()                                                                                         \
    {                                                                                                               \
        const int d0_ = f_tdi * TD, h0_ = f_thi * 8, w0_ = f_twi * TW;                                              \
        const int org_ = ((f_b * p.D + d0_) * p.H + h0_) * p.W + w0_;                                               \
        fl_bx = org_ * p.in_ld * 2;                                                                                 \
        fl_bdy = org_ * p.dy_ld * 2;                                                                                \
        fl_flags = 64 | (d0_ == 0 ? 1 : 0) | (d0_ + TD == p.D ? 2 : 0) | (h0_ == 0 ? 4 : 0) |                       \
                   (h0_ + 8 == p.H ? 8 : 0) | (w0_ == 0 ? 16 : 0) | (w0_ + TW == p.W ? 32 : 0);                     \
        if (f_t + p.groups <= last_tile) {                                                                          \
            f_t += p.groups;                                                                                        \
            f_twi += gw_;                                                                                           \
            int c_ = f_twi >= p.nw ? 1 : 0;                                                                         \
            f_twi -= c_ ? p.nw : 0;                                                                                 \
            f_thi += gh_ + c_;                                                                                      \
            c_ = f_thi >= p.nh ? 1 : 0;                                                                             \
            f_thi -= c_ ? p.nh : 0;                                                                                 \
            f_tdi += gd_ + c_;                                                                                      \
            c_ = f_tdi >= p.nd ? 1 : 0;                                                                             \
            f_tdi -= c_ ? p.nd : 0;                                                                                 \
            f_b += gb_ + c_;                                                                                        \
        }                                                                                                           \
    }
    // piece U_ (compile-time) of the open fill -> image BUF_
#define HUPR_WG_PIECE(U_, BUF_)                                                                                     \
    if ((U_) < NP - 1 || wave_u < NLAST) {                                                                          \
        const int pc_ = (U_) * 8 + wave_u;                       /* wave-uniform */                                 \
        const bool isx_ = pc_ < PX;                                                                                 \
        const int voff_ = (msk[U_] & fl_flags) ? kOOB : (isx_ ? fl_bx : fl_bdy) + rel[U_];                          \
        const unsigned dst_ = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)(BUF_) + pc_ * 1024; \
        unsigned keep_;                                                                                             \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\t" \
                     "s_mov_b32 m0, %0"                                                                             \
                     : "=&s"(keep_)                                                                                 \
                     : "s"(dst_), "v"(voff_), "s"(isx_ ? rxs : rdys)                                                \
                     : "memory");                                                                                   \
    }

    bf16x8 a[2][4], xq[2][3];
    // K-steps KS0_, KS0_ + 1 (32 voxels each) of the staged tile IMG_ as 6 groups (K-step, ky) of three taps, fragments one group ahead.
    //   3-D tile 2 x 8 x 8:  K-step kst = (depth slice kst >> 1, row half kst & 1);  2-D tile 1 x 8 x 16: (row half kst >> 1, column half kst & 1)
#define HUPR_M16_LOAD(IMG_, SET_, KS0_, J_)                                                                         \
    {                                                                                                               \
        constexpr int kst_ = (KS0_) + (J_) / 3, ky_ = (J_) % 3;                                                     \
        constexpr int xoff_ = IS3D ? ((kst_ >> 1) * HH * HW + (4 * (kst_ & 1) + ky_) * HW)                          \
                                   : ((4 * (kst_ >> 1) + ky_) * HW + 8 * (kst_ & 1));                               \
        constexpr int dyoff_ = IS3D ? 32 * kst_ : (64 * (kst_ >> 1) + 8 * (kst_ & 1));                              \
        _Pragma("unroll") for (int kx = 0; kx < 3; ++kx)                                                            \
            xq[SET_][kx] = tr_pair((IMG_) + (xoff_ + kx) * kRowB, xa[(xoff_ + kx) & 7][0], xa[(xoff_ + kx) & 7][1]); \
        if (ky_ == 0) {                                                                                             \
            _Pragma("unroll") for (int cob = 0; cob < 4; ++cob)                                                     \
                a[kst_ & 1][cob] = tr_pair((IMG_) + dyoff_ * kRowB, dyb[0][cob], dyb[1][cob]);                      \
        }                                                                                                           \
    }
    // group J_ of NJ_: its twelve MFMAs (3 kx x 4 co blocks); under them the next group's fragments — of this tile (J_ + 1 < NJ_) or
    // group 0 of the tile in NXT_ — and, in the first groups, the pieces of the open fill -> FREE_
#define HUPR_M16_STEP(IMG_, NXT_, FREE_, KS0_, J_, NJ_)                                                             \
    {                                                                                                               \
        if ((J_) + 1 < (NJ_)) { HUPR_M16_LOAD(IMG_, ((J_) + 1) & 1, KS0_, ((J_) + 1 < (NJ_) ? (J_) + 1 : 0)) }      \
        else { HUPR_M16_LOAD(NXT_, 0, KS0_, 0) }                                                                    \
        if ((J_) + 1 < (NJ_)) {                                                                                     \
            _Pragma("unroll") for (int u_ = 0; u_ < NP; ++u_)                                                       \
                if (u_ % ((NJ_) - 1) == (J_)) { HUPR_WG_PIECE(u_, FREE_) }                                          \
        }                                                                                                           \
        _Pragma("unroll") for (int kx = 0; kx < 3; ++kx)                                                            \
            _Pragma("unroll") for (int cob = 0; cob < 4; ++cob)                                                     \
                acc[((J_) % 3) * 3 + kx][cob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                            \
                    a[((KS0_) + (J_) / 3) & 1][cob], xq[(J_) & 1][kx], acc[((J_) % 3) * 3 + kx][cob], 0, 0, 0);     \
        /* the next group's fragment reads spread between this group's MFMAs */                                     \
        _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) {                                                          \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                      \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                      \
        }                                                                                                           \
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
    }
    // all groups but the last of a tile half (6 groups)
#define HUPR_M16_HEAD6(IMG_, NXT_, FREE_, KS0_)                                                                     \
    HUPR_M16_STEP(IMG_, NXT_, FREE_, KS0_, 0, 6) HUPR_M16_STEP(IMG_, NXT_, FREE_, KS0_, 1, 6) HUPR_M16_STEP(IMG_, NXT_, FREE_, KS0_, 2, 6) \
    HUPR_M16_STEP(IMG_, NXT_, FREE_, KS0_, 3, 6) HUPR_M16_STEP(IMG_, NXT_, FREE_, KS0_, 4, 6)
    // CI32: a K quarter = K-step KS_ = 3 groups per tile.  With an odd group count the fragment sets cannot have static roles per group: tile
    // parity TP_ (compile-time; the loop is unrolled by six) picks them — group J_ multiplies xq[(TP_ + J_) & 1] and the tile's dy
    // fragments a[TP_]; under the last group the next tile's first fragments land in the OTHER sets.
#define HUPR_M16Q_LOAD(IMG_, XSET_, ASET_, KS_, KY_)                                                                \
    {                                                                                                               \
        constexpr int xoff_ = IS3D ? (((KS_) >> 1) * HH * HW + (4 * ((KS_) & 1) + (KY_)) * HW)                      \
                                   : ((4 * ((KS_) >> 1) + (KY_)) * HW + 8 * ((KS_) & 1));                           \
        constexpr int dyoff_ = IS3D ? 32 * (KS_) : (64 * ((KS_) >> 1) + 8 * ((KS_) & 1));                           \
        _Pragma("unroll") for (int kx = 0; kx < 3; ++kx)                                                            \
            xq[XSET_][kx] = tr_pair((IMG_) + (xoff_ + kx) * kRowB, xa[(xoff_ + kx) & 7][0], xa[(xoff_ + kx) & 7][1]); \
        if ((KY_) == 0) {                                                                                           \
            _Pragma("unroll") for (int cob = 0; cob < 4; ++cob)                                                     \
                a[ASET_][cob] = tr_pair((IMG_) + dyoff_ * kRowB, dyb[0][cob], dyb[1][cob]);                         \
        }                                                                                                           \
    }
#define HUPR_M16Q_STEP(IMG_, NXT_, FREE_, KS_, J_, TP_)                                                             \
    {                                                                                                               \
        if ((J_) < 2) { HUPR_M16Q_LOAD(IMG_, ((TP_) + (J_) + 1) & 1, TP_, KS_, ((J_) < 2 ? (J_) + 1 : 1)) }         \
        else { HUPR_M16Q_LOAD(NXT_, ((TP_) + 1) & 1, (TP_) ^ 1, KS_, 0) }                                           \
        if ((J_) < 2) {                                                                                             \
            _Pragma("unroll") for (int u_ = 0; u_ < NP; ++u_)                                                       \
                if (u_ % 2 == (J_)) { HUPR_WG_PIECE(u_, FREE_) }                                                    \
        }                                                                                                           \
        _Pragma("unroll") for (int kx = 0; kx < 3; ++kx)                                                            \
            _Pragma("unroll") for (int cob = 0; cob < 4; ++cob)                                                     \
                acc[(J_) * 3 + kx][cob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                  \
                    a[TP_][cob], xq[((TP_) + (J_)) & 1][kx], acc[(J_) * 3 + kx][cob], 0, 0, 0);                     \
        _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) {                                                          \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                      \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                      \
        }                                                                                                           \
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                          \
    }
#define HUPR_M16_HEAD3(IMG_, NXT_, FREE_, KS_, TP_)                                                                 \
    HUPR_M16Q_STEP(IMG_, NXT_, FREE_, KS_, 0, TP_) HUPR_M16Q_STEP(IMG_, NXT_, FREE_, KS_, 1, TP_)

    // Tile protocol (round 5; the SQ counters of the old one — barrier, whole fill, first fragments, then the MFMAs — showed the
    // matrix pipe busy 53 % of the cycles: profiles/r04_wgrad_sq_pmc.txt).  Three images in a ring: CUR (tile st), NXT (tile
    // st + 1, landing or landed), FREE (tile st - 1's, released by the previous barrier).  A wave multiplies all groups of tile st
    // but the last, issuing one piece of the fill of tile st + 2 -> FREE under each of the first groups; then it waits until its
    // own pieces of tile st + 1 have landed (all but the youngest NP or NP - 1 operations: those of tile st + 2) and its reads of
    // CUR have returned, and takes the tile's ONE barrier holding the last group's fragments: it leaves the barrier with three
    // MFMAs ready and reads the first fragments of tile st + 1 under them.  The loop starts two (virtual) tiles early with the
    // multiply switched off.
#define HUPR_WG_ITER(CUR_, NXT_, FREE_, TP_)                                                                        \
    {                                                                                                               \
        if (st >= p.n_spatial) break;                                                                               \
        HUPR_WG_FILL_OPEN()                                                                                         \
        if (st >= 0) {                                                                                              \
            if constexpr (CI32) {                                                                                   \
                HUPR_M16_HEAD3(CUR_, NXT_, FREE_, 0, TP_)       /* (the K quarter's offset sits in the address tables) */ \
            } else {                                                                                                \
                if (kq == 0) { HUPR_M16_HEAD6(CUR_, NXT_, FREE_, 0) } else { HUPR_M16_HEAD6(CUR_, NXT_, FREE_, 2) } \
            }                                                                                                       \
        } else {                                                                                                    \
            _Pragma("unroll") for (int u_ = 0; u_ < NP; ++u_) { HUPR_WG_PIECE(u_, FREE_) }                          \
        }                                                                                                           \
        if (wave_u < NLAST) __builtin_amdgcn_s_waitcnt(0x0070 | NP);                                                \
        else __builtin_amdgcn_s_waitcnt(0x0070 | (NP - 1));                                                         \
        __builtin_amdgcn_s_barrier();                                                                               \
        asm volatile("" ::: "memory");                                                                              \
        if constexpr (CI32) {                                                                                       \
            if (st >= 0) {                                                                                          \
                HUPR_M16Q_STEP(CUR_, NXT_, FREE_, 0, 2, TP_)                                                        \
            } else if (st + p.groups >= 0) {                                                                        \
                HUPR_M16Q_LOAD(NXT_, ((TP_) + 1) & 1, (TP_) ^ 1, 0, 0)                                              \
            }                                                                                                       \
        } else if (st >= 0) {                                                                                       \
            if (kq == 0) { HUPR_M16_STEP(CUR_, NXT_, FREE_, 0, 5, 6) } else { HUPR_M16_STEP(CUR_, NXT_, FREE_, 2, 5, 6) } \
        } else if (st + p.groups >= 0) {                        /* the first real tile's first fragments */         \
            if (kq == 0) { HUPR_M16_LOAD(NXT_, 0, 0, 0) } else { HUPR_M16_LOAD(NXT_, 0, 2, 0) }                     \
        }                                                                                                           \
        st += p.groups;                                                                                             \
    }
    static_assert(NP < 16 && NP <= 11, "piece count exceeds the counted wait / the groups of a tile");
    if (group >= p.n_spatial) return;
    int st = group - 2 * p.groups;
    for (;;) {
        HUPR_WG_ITER(bufB, bufC, bufA, 0)
        HUPR_WG_ITER(bufC, bufA, bufB, 1)
        HUPR_WG_ITER(bufA, bufB, bufC, 0)
        if constexpr (CI32) {                        // (tile parity: see HUPR_M16Q_STEP; the other form ignores it)
            HUPR_WG_ITER(bufB, bufC, bufA, 1)
            HUPR_WG_ITER(bufC, bufA, bufB, 0)
            HUPR_WG_ITER(bufA, bufB, bufC, 1)
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);              // the two re-fetches past the last tile must land before the LDS is released
#undef HUPR_WG_ITER
#undef HUPR_WG_FILL_OPEN
#undef HUPR_WG_PIECE
#undef HUPR_M16Q_LOAD
#undef HUPR_M16Q_STEP
#undef HUPR_M16_LOAD
#undef HUPR_M16_STEP
#undef HUPR_M16_HEAD6
#undef HUPR_M16_HEAD3

    // Merge the two K halves through the (now dead) images: the kq = 1 waves park their accumulators, six taps and then three, the
    // kq = 0 waves add them (both halves hold the same (co, ci) element in the same lane and register) — one partial tensor per
    // workgroup.
    if constexpr (CI32) {
        // four K quarters: quarters 1-3 park three taps at a time (one image per tap), quarter 0 adds them in the order 1, 2, 3
        const int t128 = tid & 127;
        float* const red[3] = {reinterpret_cast<float*>(bufA), reinterpret_cast<float*>(bufB), reinterpret_cast<float*>(bufC)};
        static_assert(3 * 16 * 128 * 4 <= IMG, "an image must hold one tap of three quarters");
        __syncthreads();
#pragma unroll
        for (int round = 0; round < 3; ++round) {
            if (round) __syncthreads();
            if (kq != 0) {
#pragma unroll
                for (int tl = 0; tl < 3; ++tl)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[tl][(((kq - 1) * 16 + r) << 7) + t128] = acc[3 * round + tl][r >> 2][r & 3];
            }
            __syncthreads();
            if (kq == 0) {
#pragma unroll
                for (int tl = 0; tl < 3; ++tl)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
#pragma unroll
                        for (int q = 0; q < 3; ++q) acc[3 * round + tl][r >> 2][r & 3] += red[tl][((q * 16 + r) << 7) + t128];
            }
        }
    } else {
        const int t256 = tid & 255;
        float* const red[3] = {reinterpret_cast<float*>(bufA), reinterpret_cast<float*>(bufB), reinterpret_cast<float*>(bufC)};
        __syncthreads();
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            if (round) __syncthreads();
            if (kq == 1) {
#pragma unroll
                for (int tap = round * 6; tap < (round ? 9 : 6); ++tap)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((tap - round * 6) >> 1)][(((tap & 1) * 16 + r) << 8) + t256] = acc[tap][r >> 2][r & 3];
            }
            __syncthreads();
            if (kq == 0) {
#pragma unroll
                for (int tap = round * 6; tap < (round ? 9 : 6); ++tap)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tap][r >> 2][r & 3] += red[((tap - round * 6) >> 1)][(((tap & 1) * 16 + r) << 8) + t256];
            }
        }
    }
    if (kq != 0) return;
    const int ci = ci0 + 16 * cb + s;
    float* part = p.part + (long)group * p.Co * T * p.Ci;
    if (ci < p.Ci) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int cob = 0; cob < 4; ++cob)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int co = co0 + 16 * cob + 4 * kq4 + i;
                    if (co < p.Co) part[((long)co * T + td * 9 + tap) * p.Ci + ci] = acc[tap][cob][i];
                }
        }
    }
}

}  // namespace hupr

using namespace hupr;

extern "C" size_t hupr_conv3x3_wgrad_halo_ws_bytes(int Ci, int Co, int kd) {
    // at most 256 partial tensors, but never more than 128 MiB of them
    const size_t one = (size_t)Co * kd * 9 * Ci * sizeof(float);
    size_t groups = 256;
    while (groups > 1 && groups * one > ((size_t)128 << 20)) groups >>= 1;
    return groups * one;
}

static int g_wgrad_m16 = 1;         // A/B aid (hupr_debug_wgrad_m16): 0 = the 32 x 32 x 16 kernel (rounds 2-4)
extern "C" void hupr_debug_wgrad_m16(int on) { g_wgrad_m16 = on; }
static int g_wgrad_groups256 = 1;   // (hupr_debug_wgrad_ci32(16 + mode): 128 partial tensors at most, as before)
static int g_wgrad_ci32 = 1;        // A/B aid (hupr_debug_wgrad_ci32): 0 = Ci <= 32 through the two-quadrant kernel as before, 2 = K quarters always
extern "C" void hupr_debug_wgrad_ci32(int on) { g_wgrad_ci32 = on & 15; g_wgrad_groups256 = !(on & 16); }

// dy2 / dw2 (both or neither): a second gradient tensor of the same shape and stride over the same x — Co is then the channel count of
// EACH; one launch of the 16 x 16 x 32 kernel over 2 Co output channels and one reduction that splits its rows between dw and dw2.
// Same partial tensors and the same sums per element as two calls (the workgroup count per (depth tap, tile pair) is the single
// call's).  LDS-DMA kernels (bf16 storage) only: HUPR_ERR_ARG otherwise (the caller makes two calls).
static int wgrad_halo(const void* x, const void* dy, float* dw, int Bn, int D, int H, int W, int Ci, int in_ld, int Co,
                      int dy_ld, int kd, void* ws, size_t ws_bytes, bool abf, hupr_stream_t stream, const char* who,
                      const void* dy2 = nullptr, float* dw2 = nullptr) {
    HUPR_REQUIRE(x && dy && dw && ws, "%s: null pointer", who);
    HUPR_REQUIRE((dy2 == nullptr) == (dw2 == nullptr), "%s: dy2 and dw2 go together", who);
    const bool dual = dy2 != nullptr;
    HUPR_REQUIRE(!dual || (abf && Co % 64 == 0), "%s: the two-gradient form needs bf16 storage and Co %% 64 == 0", who);
    const int al = abf ? 8 : 4;
    HUPR_REQUIRE(Bn > 0 && Co > 0 && Ci % 8 == 0 && Co % 8 == 0 && in_ld % al == 0 && dy_ld % al == 0,
                 "%s: unsupported channels Ci=%d Co=%d", who, Ci, Co);
    HUPR_REQUIRE(H % 8 == 0 && ((kd == 3 && D % 2 == 0 && W % 8 == 0) || (kd == 1 && D == 1 && W % 16 == 0)),
                 "%s: unsupported geometry", who);
    HUPR_REQUIRE((long)Bn * D * H * W * (in_ld > dy_ld ? in_ld : dy_ld) < (1L << 31), "%s: tensor too large for 32-bit offsets", who);
    WgradHaloArgs a;
    a.xcd_map = 0;
    a.x = x; a.dy = dy; a.part = reinterpret_cast<float*>(ws);
    a.Bn = Bn; a.D = D; a.H = H; a.W = W; a.Ci = Ci; a.in_ld = in_ld; a.Co = Co; a.dy_ld = dy_ld;
    a.kd = kd;
    if (kd == 3) { a.TD = 2; a.log2TW = 3; } else { a.TD = 1; a.log2TW = 4; }
    a.nd = D / a.TD; a.nh = H / 8; a.nw = W >> a.log2TW;
    a.dy2 = dy2;
    a.co_split = dual ? Co : 0;
    a.n_ci_tiles = (Ci + 63) / 64;
    a.n_co_tiles = (Co + 63) / 64;
    a.n_spatial = Bn * a.nd * a.nh * a.nw;
    const int pairs = a.n_ci_tiles * a.n_co_tiles * kd;                 // of ONE gradient: decides the partial-tensor count below
    const int nt = dual ? 2 : 1;
    const size_t one = (size_t)nt * Co * kd * 9 * Ci * sizeof(float);
    // ~3 workgroups per CU, at most 128 partial tensors — 256 for the register-staged kernel on 2-D maps (round 6: the fp32-stored last
    // decoder block, one (co, ci) tile pair: 128 workgroups of synchronous fills left half the chip idle, 56 us for 50 MB)
    int groups = max(1, min(kd == 1 && g_wgrad_groups256 ? 256 : 128, 768 / pairs));
    groups = min(groups, a.n_spatial);
    while (groups > 1 && (size_t)groups * one > ws_bytes) groups >>= 1;
    if ((size_t)groups * one > ws_bytes) return fail(HUPR_ERR_WORKSPACE, "hupr_conv3x3_wgrad_halo_bf16: workspace too small");
    hipStream_t s = as_stream(stream);
    const long n = (long)Co * kd * 9 * Ci;
    const long max_bytes = (long)Bn * D * H * W * (in_ld > dy_ld ? in_ld : dy_ld) * 2;
    if (abf && max_bytes < 0x7ffffff0L) {
        // LDS-DMA kernel: one 512-thread workgroup per CU (two K halves, merged in LDS), one partial tensor per workgroup
        int gw = max(1, min(128, 256 / pairs));
        // XCD affinity: 8 * floor(32 / members) groups put every member of a group on one XCD with <= 32 workgroups per XCD
        const int g8 = pairs <= 32 ? 8 * (32 / pairs) : 0;
        a.xcd_map = g8 > 0 && 10 * g8 >= 9 * gw && g8 <= a.n_spatial;
        if (a.xcd_map) gw = min(g8, 128);
        gw = min(gw, a.n_spatial);
        while (gw > 1 && (size_t)gw * one > ws_bytes) { gw >>= 1; a.xcd_map = 0; }
        if (dual && (size_t)gw * one > ws_bytes) return fail(HUPR_ERR_WORKSPACE, "%s: workspace too small for two gradients", who);
        if ((size_t)gw * one <= ws_bytes) {
            a.groups = gw;
            if (dual) { a.Co = 2 * Co; a.n_co_tiles *= 2; }
            const dim3 grid = a.xcd_map ? dim3(gw * pairs * nt) : dim3(gw, kd, a.n_ci_tiles * a.n_co_tiles);
            // K quarters pay once a workgroup multiplies enough tiles to amortise the extra LDS merge (measured: 222 -> 160 us on
            // the 32 -> 64 layer-1 shape at 102 tiles per workgroup; +2-3 us on shapes with one or two tiles per workgroup)
            const bool ci32 = Ci <= 32 && (g_wgrad_ci32 >= 2 || (g_wgrad_ci32 == 1 && a.n_spatial >= 16 * gw));
            if (g_wgrad_m16 && !ci32) {                                  // the 16 x 16 x 32 form (round 5)
                if (kd == 3) HUPR_LAUNCH((hupr_k_wgrad_halo_m16<true>), grid, dim3(512), 0, s, a);
                else HUPR_LAUNCH((hupr_k_wgrad_halo_m16<false>), grid, dim3(512), 0, s, a);
            } else if (g_wgrad_m16 && kd == 3 && g_wgrad_ci32 != 3) {    // ... and its K-quarter form for Ci <= 32 (round 6; hupr_debug_wgrad_ci32(3): the 32 x 32 x 16 one)
                HUPR_LAUNCH((hupr_k_wgrad_halo_m16<true, true>), grid, dim3(512), 0, s, a);
            } else if (kd == 3) {
                if (ci32) HUPR_LAUNCH((hupr_k_wgrad_halo_glds<true, true>), grid, dim3(512), 0, s, a);
                else HUPR_LAUNCH((hupr_k_wgrad_halo_glds<true, false>), grid, dim3(512), 0, s, a);
            } else {
                if (ci32) HUPR_LAUNCH((hupr_k_wgrad_halo_glds<false, true>), grid, dim3(512), 0, s, a);
                else HUPR_LAUNCH((hupr_k_wgrad_halo_glds<false, false>), grid, dim3(512), 0, s, a);
            }
            HUPR_LAUNCH_OK("hupr_k_wgrad_halo_glds");
            launch_splitk_reduce(reinterpret_cast<const float*>(ws), dw, nt * n, gw, nt * n, kd * 9, Ci, s, dw2, n);
            HUPR_LAUNCH_OK("hupr_k_splitk_reduce");
            return HUPR_OK;
        }
    }
    if (dual) return fail(HUPR_ERR_ARG, "%s: the two-gradient form applies to the LDS-DMA kernel's envelope only", who);
    a.groups = groups;
    const dim3 grid(groups, kd, a.n_ci_tiles * a.n_co_tiles);
    if (kd == 3) {
        if (abf) HUPR_LAUNCH((hupr_k_wgrad_halo_bf16<true, true>), grid, dim3(256), 0, s, a);
        else HUPR_LAUNCH((hupr_k_wgrad_halo_bf16<false, true>), grid, dim3(256), 0, s, a);
    } else {
        if (abf) HUPR_LAUNCH((hupr_k_wgrad_halo_bf16<true, false>), grid, dim3(256), 0, s, a);
        else HUPR_LAUNCH((hupr_k_wgrad_halo_bf16<false, false>), grid, dim3(256), 0, s, a);
    }
    HUPR_LAUNCH_OK("hupr_k_wgrad_halo_bf16");
    launch_splitk_reduce(reinterpret_cast<const float*>(ws), dw, n, groups, n, kd * 9, Ci, s);
    HUPR_LAUNCH_OK("hupr_k_splitk_reduce");
    return HUPR_OK;
}

extern "C" int hupr_conv3x3_wgrad_halo_bf16(const float* x, const float* dy, float* dw, int Bn, int D, int H, int W, int Ci,
                                            int in_ld, int Co, int dy_ld, int kd, void* ws, size_t ws_bytes,
                                            hupr_stream_t stream) {
    return wgrad_halo(x, dy, dw, Bn, D, H, W, Ci, in_ld, Co, dy_ld, kd, ws, ws_bytes, false, stream,
                      "hupr_conv3x3_wgrad_halo_bf16");
}

// x and dy stored as bf16 (leading dimensions in elements); dw stays fp32 in parameter layout.
// two weight gradients over the same x (the two convolutions of a residual block): see wgrad_halo.  ws: twice
// hupr_conv3x3_wgrad_halo_ws_bytes(Ci, Co, kd) is always enough.  hupr_conv3x3_wgrad_halo_dual_supported: 1 where this call applies.
extern "C" int hupr_conv3x3_wgrad_halo_dual_supported(int Bn, int D, int H, int W, int Ci, int Co, int kd) {
    if (!(Bn > 0 && Ci % 8 == 0 && Co % 64 == 0 && H % 8 == 0 &&
          ((kd == 3 && D % 2 == 0 && W % 8 == 0) || (kd == 1 && D == 1 && W % 16 == 0))))
        return 0;
    return (long)Bn * D * H * W * (Ci > Co ? Ci : Co) * 2 < 0x7ffffff0L ? 1 : 0;
}
extern "C" int hupr_conv3x3_wgrad_halo_bf16act_dual(const void* x, const void* dy_a, const void* dy_b, float* dw_a, float* dw_b, int Bn,
                                                    int D, int H, int W, int Ci, int in_ld, int Co, int dy_ld, int kd, void* ws,
                                                    size_t ws_bytes, hupr_stream_t stream) {
    HUPR_REQUIRE(dy_b && dw_b, "hupr_conv3x3_wgrad_halo_bf16act_dual: null pointer");
    return wgrad_halo(x, dy_a, dw_a, Bn, D, H, W, Ci, in_ld, Co, dy_ld, kd, ws, ws_bytes, true, stream,
                      "hupr_conv3x3_wgrad_halo_bf16act_dual", dy_b, dw_b);
}

extern "C" int hupr_conv3x3_wgrad_halo_bf16act(const void* x, const void* dy, float* dw, int Bn, int D, int H, int W,
                                               int Ci, int in_ld, int Co, int dy_ld, int kd, void* ws, size_t ws_bytes,
                                               hupr_stream_t stream) {
    return wgrad_halo(x, dy, dw, Bn, D, H, W, Ci, in_ld, Co, dy_ld, kd, ws, ws_bytes, true, stream,
                      "hupr_conv3x3_wgrad_halo_bf16act");
}
