// Fused (flash-style) MSCSA attention on the bf16 matrix pipe — no N x N matrix ever reaches HBM.
//
// Reference semantics (models/layers.py:126-133), token-major fp32 tensors (B, N, C):
//     S[j,q] = sum_c K[j,c] Q[q,c]     P = softmax over keys j     out[q,c] = sum_j P[j,q] V[j,c] (+ V[q,c])
// i.e. standard single-head attention with scale 1; C = 64 or 128, N % 128 == 0.
//
// All tiles are computed in the "keys x queries" orientation S^T = K Q^T so that one lane owns one query
// column: the online-softmax statistics (running max / sum, LSE, D = rowsum(dO o O)) are lane-local
// scalars and the only cross-lane step is one exchange between the two half-waves.  A probability tile
// in its MFMA accumulator layout is ALREADY a valid B operand for the next MFMA (K slot 8h+i <-> key
// 16u + 8(i>>2) + 4h + (i&3)); the matching A operand (V^T, K^T, dO^T, Q^T: rows = channels, K = tokens)
// is produced from the row-major LDS image by ds_read_b64_tr_b16 with exactly those rows.
//
//   forward : per 128-query workgroup (32 per wave) stream 64-key tiles: S^T, online softmax, O^T += V^T P^T
//   backward: dQ kernel (same walk): P^T = exp(S^T - LSE), dP^T = V dO^T, dS^T = P^T o (dP^T - D),
//             dQ^T += K^T dS^T;   dK/dV kernel (per 128-key workgroup, stream 64-query tiles):
//             S = Q K^T, P, dP = dO V^T, dS;  dV^T += dO^T P,  dK^T += Q^T dS   (+ dO for the residual)
#include "gemm_common.h"

namespace hupr {

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
// QS kernels (round 5, "query pre-scaled"): the caller hands over Q' = log2(e) . Q (the factor folded into the 1x1 query
// projection before its bf16 rounding, functional.MSCSALevelFn), so that K . Q'^T is the exponent of 2 directly; the MFMA chain
// of a score tile then STARTS from minus the running maximum (forward) or minus the stored log-sum-exp (backward) as its
// accumulator input, and the score leaves the matrix pipe as the argument of v_exp_f32 — the per-score fma of rounds 1-4
// (62 of ~276 VALU instructions per two key tiles of the forward, profiles/r04b_attn_isa_loop_mix.txt) is gone.  The forward
// keeps the running maximum it started a tile with unless the tile exceeds it by more than kDeferBits binary orders (the rescale
// branch, which was conditional already): P then lies in (0, 2^kDeferBits] instead of (0, 1] at the same relative bf16
// precision; row sum and O scale with it — the same soft-max.
constexpr float kDeferBits = 8.f;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int D>
struct Img {   // row-major bf16 LDS image [rows][D] with XOR-swizzled 16-byte chunks
    static constexpr int CH = D / 8;
    // key(row): which 16-byte chunk position a row's chunk c lands on (c ^ key).  Two kinds of reads hit an image: ds_read_b128 of
    // one chunk position over 16 rows per lane group (rows {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} + 32), and ds_read_b64_tr_b16
    // of four consecutive rows x one 64-byte column segment per 32-lane group.  Rounds 1-4 used the row bits in place
    // ((row >> 1) & 7 at D = 64, row & 15 / 31 above): conflict-free for the first kind, but the four rows of a transpose read then
    // differ only in the LOW key bits, which permute chunks inside the same 64-byte segment — rows r and r + 2 (D = 64; all four
    // rows at D >= 128) meet on the same 16 banks: every transpose read took 2 (4) LDS cycles per group, 33 / 20 / 22 % of all LDS
    // cycles of the forward / dQ / dK-dV kernels (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, profiles/r04b_attn_sq_pmc.txt; the
    // model in scripts/lds_bank_model.py gives the same 33 %).  Rotating the key so that the fastest-changing row bits select the
    // 64-byte segment keeps it a bijection on every b128 lane group and separates the rows of a transpose read: 0 conflict cycles
    // in the model for both kinds at D = 64, 128, 256.
    static __device__ __forceinline__ int key(int row) {
        return (D == 64) ? ((((row >> 1) & 1) << 2) | ((row >> 2) & 3)) : ((((row & 3) << 2) | ((row >> 2) & 3)) | (row & (D / 8 - 1) & ~15));
    }
    static __device__ __forceinline__ int off(int row, int chunk) { return row * D + ((chunk ^ key(row)) << 3); }   // bf16 elements
};

typedef unsigned u32x4a __attribute__((ext_vector_type(4)));

// stage ROWS x D rows (fp32, rounded here, or already-bf16; row stride ld elements) into a swizzled bf16 image
template <int D, int ROWS, typename TI>
__device__ __forceinline__ void stage_rows(__bf16* img, const TI* __restrict__ src, long ld, int tid) {
    constexpr int CH = D / 8, ITEMS = ROWS * CH, PER = ITEMS / 256;
    static_assert(ITEMS % 256 == 0, "tile must split over 256 threads");
    if constexpr (sizeof(TI) == 2) {
        u32x4a a[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int it = tid + 256 * i, row = it / CH, ch = it % CH;
            a[i] = *reinterpret_cast<const u32x4a*>(src + (long)row * ld + ch * 8);
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int it = tid + 256 * i, row = it / CH, ch = it % CH;
            *reinterpret_cast<u32x4a*>(&img[Img<D>::off(row, ch)]) = a[i];
        }
    } else {
        float4 a[PER], c[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int it = tid + 256 * i, row = it / CH, ch = it % CH;
            const float* s = reinterpret_cast<const float*>(src) + (long)row * ld + ch * 8;
            a[i] = *reinterpret_cast<const float4*>(s);
            c[i] = *reinterpret_cast<const float4*>(s + 4);
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int it = tid + 256 * i, row = it / CH, ch = it % CH;
            bf16x8 v;
            v[0] = (__bf16)a[i].x; v[1] = (__bf16)a[i].y; v[2] = (__bf16)a[i].z; v[3] = (__bf16)a[i].w;
            v[4] = (__bf16)c[i].x; v[5] = (__bf16)c[i].y; v[6] = (__bf16)c[i].z; v[7] = (__bf16)c[i].w;
            *reinterpret_cast<bf16x8*>(&img[Img<D>::off(row, ch)]) = v;
        }
    }
}

// The same staging split in two so that the global loads of tile i + 1 are in flight while tile i is multiplied:
// load() right after the tile barrier, store() after the next one.
template <int D, int ROWS, typename TI>
struct StageRegs {
    static constexpr int CH = D / 8, ITEMS = ROWS * CH, PER = ITEMS / 256;
    static_assert(ITEMS % 256 == 0, "tile must split over 256 threads");
    u32x4a v[PER * (sizeof(TI) == 2 ? 1 : 2)];
    __device__ __forceinline__ void load(const TI* __restrict__ src, long ld, int tid) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int it = tid + 256 * i, row = it / CH, ch = it % CH;
            if constexpr (sizeof(TI) == 2) {
                v[i] = *reinterpret_cast<const u32x4a*>(src + (long)row * ld + ch * 8);
            } else {
                const float* sp = reinterpret_cast<const float*>(src) + (long)row * ld + ch * 8;
                v[2 * i] = *reinterpret_cast<const u32x4a*>(sp);
                v[2 * i + 1] = *reinterpret_cast<const u32x4a*>(sp + 4);
            }
        }
    }
    __device__ __forceinline__ void store(__bf16* img, int tid) const {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int it = tid + 256 * i, row = it / CH, ch = it % CH;
            if constexpr (sizeof(TI) == 2) {
                *reinterpret_cast<u32x4a*>(&img[Img<D>::off(row, ch)]) = v[i];
            } else {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (__bf16)__uint_as_float(v[2 * i][e]);
                    o[4 + e] = (__bf16)__uint_as_float(v[2 * i + 1][e]);
                }
                *reinterpret_cast<bf16x8*>(&img[Img<D>::off(row, ch)]) = o;
            }
        }
    }
};

// B-operand fragments (cols = token of this lane, K = channels) straight from global: frag[ks] covers
// channels 16 ks + 8 h .. +7 of row `tok`
template <int D, typename TI>
__device__ __forceinline__ void load_frags(bf16x8* frag, const TI* __restrict__ rowp, int lh) {
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) {
        if constexpr (sizeof(TI) == 2) {
            frag[ks] = *reinterpret_cast<const bf16x8*>(rowp + ks * 16 + lh * 8);
        } else {
            const float* rp = reinterpret_cast<const float*>(rowp);
            const float4 a = *reinterpret_cast<const float4*>(rp + ks * 16 + lh * 8);
            const float4 c = *reinterpret_cast<const float4*>(rp + ks * 16 + lh * 8 + 4);
            bf16x8 v;
            v[0] = (__bf16)a.x; v[1] = (__bf16)a.y; v[2] = (__bf16)a.z; v[3] = (__bf16)a.w;
            v[4] = (__bf16)c.x; v[5] = (__bf16)c.y; v[6] = (__bf16)c.z; v[7] = (__bf16)c.w;
            frag[ks] = v;
        }
    }
}

// acc[t] (t = 0,1: image rows 32t..32t+31) = img(64 rows x D) . frags  ->  tile [image row][lane token]
constexpr int g_attn_sched = 0;      // 1: the round-1 order (all fragment reads of a chunk, then its MFMAs)
// ZERO = false: acc arrives initialised (the QS kernels start the chain from -max / -log-sum-exp)
// NT = image rows / 32 (2; 1: the 32-key tiles of the level-2 dQ kernel)
template <int D, bool ZERO = true, int NT = 2>
__device__ __forceinline__ void mma_rows_x_frags(f32x16* acc, const __bf16* img, const bf16x8* frag, int lr, int lh) {
    // all A fragments of the 64 x D image rows are read before the first MFMA (hipcc otherwise issues each read right in
    // front of its MFMA and every MFMA waits out the LDS latency)
    // (D = 256: in chunks of four K-steps — sixteen would hold 128 registers of fragments)
    constexpr int KS = D / 16, CHK = KS >= 8 ? 4 : KS;      // (D = 128: chunks of four since round 6 — 32 fragment registers instead of 64; same MFMA order)
    if constexpr (ZERO) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        }
    }
#pragma unroll
    for (int k0 = 0; k0 < KS; k0 += CHK) {
        bf16x8 a[NT][CHK];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < CHK; ++ks)                   // in the order the MFMAs consume them
#pragma unroll
            for (int t = 0; t < NT; ++t)
                a[t][ks] = *reinterpret_cast<const bf16x8*>(&img[Img<D>::off(32 * t + lr, (k0 + ks) * 2 + lh)]);
#pragma unroll
        for (int ks = 0; ks < CHK; ++ks)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][ks], frag[k0 + ks], acc[t], 0, 0, 0);
        if (g_attn_sched == 0) {
            // four reads ahead, then one read behind every MFMA (a burst of all 2 CHK reads first delays the first MFMA by the
            // whole burst; a read right in front of its MFMA exposes the LDS latency every time)
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int i_ = 0; i_ < NT * CHK - 4; ++i_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        } else {
            __builtin_amdgcn_sched_group_barrier(0x100, NT * CHK, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NT * CHK, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// out[ct] (channels 32ct..) += img^T (channels x 64 image rows) . W, where W is a [64 image rows][lane token] tile
// held in accumulator layout w[2][16] (as produced by mma_rows_x_frags).  K slot 8h+i of K-step (t,u) <-> image row
// 32t + 16u + 8(i>>2) + 4h + (i&3); the A operand rows are fetched with transpose-reads.
// NCT channel tiles starting at tile ct0 (a channel half of the dK / dV kernel at D = 256; everything otherwise).
template <int D, int NCT = D / 32, int NT = 2>
__device__ __forceinline__ void mma_tr_x_tile(f32x16* out, const __bf16* img, const f32x16* w, int lane, int ct0 = 0) {
    const int g = lane >> 4, s = lane & 15, h = g >> 1;
    const int col = 16 * (g & 1) + 4 * (s & 3);            // first of the 4 channels this supplier lane addresses
    const int rsub = s >> 2;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            bf16x8 b;
#pragma unroll
            for (int i = 0; i < 8; ++i) b[i] = (__bf16)w[t][8 * u + i];
            const int row0 = 32 * t + 16 * u + 4 * h + rsub, row1 = row0 + 8;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const int c = 32 * (ct0 + ct) + col;           // channel; chunk = c >> 3, 8-byte half = (c >> 2) & 1
                const __bf16* p0 = &img[Img<D>::off(row0, c >> 3) + (c & 4)];
                const __bf16* p1 = &img[Img<D>::off(row1, c >> 3) + (c & 4)];
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
                union { struct { s16x4 a, b; } s2; bf16x8 v; } uu;
                uu.s2.a = lo;
                uu.s2.b = hi;
                out[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uu.v, b, out[ct], 0, 0, 0);
            }
        }
    }
}

// write an accumulator tile set acc[ct] ([channel][lane token]) to dst[token][channel] (+ scale, + optional add)
template <int D>
__device__ __forceinline__ void store_ct(float* dst_row, const f32x16* acc, float scale, const float* add_row, int lh) {
    // add_row may alias dst_row (gradient accumulation in place): each float4 is read, then written, by one lane
#pragma unroll
    for (int ct = 0; ct < D / 32; ++ct) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int c = 32 * ct + 8 * q4 + 4 * lh;
            float4 v = make_float4(acc[ct][4 * q4] * scale, acc[ct][4 * q4 + 1] * scale, acc[ct][4 * q4 + 2] * scale,
                                   acc[ct][4 * q4 + 3] * scale);
            if (add_row) {
                const float4 a = *reinterpret_cast<const float4*>(add_row + c);
                v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
            }
            *reinterpret_cast<float4*>(dst_row + c) = v;
        }
    }
}

// the same tile set stored as bf16 (8-byte runs of 4 channels): the decoder's copy of an attention output
template <int D>
__device__ __forceinline__ void store_ct16(__bf16* dst_row, const f32x16* acc, float scale, const float* add_row, int lh) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int ct = 0; ct < D / 32; ++ct) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int c = 32 * ct + 8 * q4 + 4 * lh;
            float4 v = make_float4(acc[ct][4 * q4] * scale, acc[ct][4 * q4 + 1] * scale, acc[ct][4 * q4 + 2] * scale,
                                   acc[ct][4 * q4 + 3] * scale);
            if (add_row) {
                const float4 a = *reinterpret_cast<const float4*>(add_row + c);
                v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
            }
            const bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
            *reinterpret_cast<bf16x4*>(dst_row + c) = o;
        }
    }
}
// store_ct with the added tensor given as bf16 (a gradient that arrives bf16-stored)
template <int D>
__device__ __forceinline__ void store_ct_add16(float* dst_row, const f32x16* acc, const __bf16* add_row, int lh) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int ct = 0; ct < D / 32; ++ct) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int c = 32 * ct + 8 * q4 + 4 * lh;
            const bf16x4 a = *reinterpret_cast<const bf16x4*>(add_row + c);
            *reinterpret_cast<float4*>(dst_row + c) = make_float4(acc[ct][4 * q4] + (float)a[0], acc[ct][4 * q4 + 1] + (float)a[1],
                                                                  acc[ct][4 * q4 + 2] + (float)a[2], acc[ct][4 * q4 + 3] + (float)a[3]);
        }
    }
}

// (token block bx, sample by) of this workgroup in an (nx, Bn) grid.  xcd_map (Bn % 8 == 0): workgroup id -> XCD is id % 8 and every
// XCD has its own L2, so the nx workgroups of a sample — each of which streams ALL of the sample's K / V (Q / dO) — are given ids
// that are congruent mod 8: one L2 fetches the sample's operands once instead of up to eight (round 4; the ping-pong forward does
// the same with its 1-D grid).
__device__ __forceinline__ void xcd_block(int& bx, int& by, int xcd_map) {
    bx = blockIdx.x;
    by = blockIdx.y;
    if (xcd_map) {
        const int nx = gridDim.x, L = blockIdx.y * nx + blockIdx.x, idx = L >> 3;
        bx = idx % nx;
        by = (L & 7) + 8 * (idx / nx);
    }
}

// ------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------
// TI = float (operands rounded to bf16 while staged) or __bf16 (pre-rounded copies: every workgroup re-reads all of K
// and V, so halving those bytes and dropping the per-tile conversions is worth one cast pass); Vres = fp32 V for the
// residual epilogue (exact), or null.
// SPLIT (few workgroups: small batches, e.g. the B = 1 inference of BASELINE config C2): blockIdx.z walks only its share of the
// keys and leaves the un-normalised O^T tile, the running maximum and the running sum in part_o / part_ml
// ([split][B N][D] / [split][B N][2]); hupr_k_attn_combine merges the shares (flash-decoding).
// Up to four independent attentions of equal shape in ONE launch (the four of an MSCSA level): n > 0; SPLIT (single-sample inference,
// where launches, not work, set the time): blockIdx.z = item * splits + split; one-pass kernel (training batches at the levels whose
// own grid leaves most of the chip idle — level 3: N = 256, two workgroups per sample): blockIdx.z = item.
struct AttnBatch {
    int n, splits;
    const void* K[4];
    const void* Q[4];
    const void* V[4];
    const float* Vres[4];
    float* out[4];
    float* lse[4];
    __bf16* out16[4];
};

// Two workgroups per CU (<= 256 unified registers) at D = 128 as well (round 6): the level-2 forward had compiled to 316 registers = ONE
// wave per SIMD, every MFMA result shuttled through AGPRs, nothing to overlap a wave's own latencies with (150 us for 69 GF).
template <int D, typename TI, bool SPLIT = false, bool QS = false>
__global__ __launch_bounds__(256, D <= 128 ? 2 : 1) void hupr_k_attn_fwd(const TI* __restrict__ K, const TI* __restrict__ Q,
                                                       const TI* __restrict__ V, const float* __restrict__ Vres,
                                                       float* __restrict__ out, float* __restrict__ lse, int N, int ldk,
                                                       int ldq, __bf16* __restrict__ out16, int ld16,
                                                       float* __restrict__ part_o = nullptr, float* __restrict__ part_ml = nullptr,
                                                       const AttnBatch batch = AttnBatch()) {
    // ldk / ldq: row strides (elements) of K and Q — the projections of one map may sit side by side in one tensor;
    // out16 (optional): a bf16 copy of the output with row stride ld16 (a column block of the decoder's input)
    __shared__ __attribute__((aligned(16))) __bf16 Ks[64 * D];
    __shared__ __attribute__((aligned(16))) __bf16 Vs[64 * D];
    int zs = SPLIT ? (int)blockIdx.z : 0, nzs = SPLIT ? (int)gridDim.z : 1;      // this workgroup's key share and their number
    if constexpr (SPLIT) {
        if (batch.n > 0) {
            const int item = (int)blockIdx.z / batch.splits;
            nzs = batch.splits;
            zs = (int)blockIdx.z - item * nzs;
            K = static_cast<const TI*>(batch.K[item]);
            Q = static_cast<const TI*>(batch.Q[item]);
            V = static_cast<const TI*>(batch.V[item]);
            const long rows = (long)gridDim.y * N;                               // partials: [item][split][B N]
            part_o += (long)item * nzs * rows * D;
            part_ml += (long)item * nzs * rows * 2;
        }
    }
    if constexpr (!SPLIT) {
        if (batch.n > 0) {
            const int item = (int)blockIdx.z;
            K = static_cast<const TI*>(batch.K[item]);
            Q = static_cast<const TI*>(batch.Q[item]);
            V = static_cast<const TI*>(batch.V[item]);
            Vres = batch.Vres[item];
            out = batch.out[item];
            lse = batch.lse[item];
            out16 = batch.out16[item];
        }
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
    // (query block bx, sample by): with Bn % 8 == 0 the query blocks of one sample are given workgroup ids that are congruent mod 8 =
    // one XCD, whose L2 then fetches the sample's K / V once (round 6: in plain grid order the eight query blocks of a level-2 sample
    // sat on eight different XCDs — every workgroup streams ALL keys and values of its sample and lives only ~9 us: the SQ counters
    // showed one resident wave per CU on average in a 150 us launch, profiles/r06_attn_l2_sq_pmc.txt; the backward kernels and the
    // ping-pong forward have mapped their grids this way since round 4)
    int bx, by;
    xcd_block(bx, by, (!SPLIT && gridDim.y % 8 == 0) ? 1 : 0);
    const long base = (long)by * N * D;
    const int q = bx * 128 + wave * 32 + lr;                 // this lane's query
    K += (long)by * N * ldk;
    bf16x8 qf[D / 16];
    load_frags<D, TI>(qf, Q + ((long)by * N + q) * ldq, lh);
    f32x16 o[D / 32];
#pragma unroll
    for (int ct = 0; ct < D / 32; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    constexpr bool PF = (D <= 128);                           // D = 256 has no registers to spare for the prefetch
    StageRegs<PF ? D : 64, 64, TI> kr, vr;
    const int jb = SPLIT ? zs * (N / nzs) : 0;                              // this workgroup's key range [jb, je)
    const int je = SPLIT ? jb + N / nzs : N;
    if (PF) {
        kr.load(K + (long)jb * ldk, ldk, tid);
        vr.load(V + base + (long)jb * D, D, tid);
    }
    for (int j0 = jb; j0 < je; j0 += 64) {
        __syncthreads();
        if (PF) {
            kr.store(Ks, tid);
            vr.store(Vs, tid);
        } else {
            stage_rows<D, 64, TI>(Ks, K + (long)j0 * ldk, ldk, tid);
            stage_rows<D, 64, TI>(Vs, V + base + (long)j0 * D, D, tid);
        }
        __syncthreads();
        if (PF && j0 + 64 < je) {                             // next tile's rows travel while this one is multiplied
            kr.load(K + (long)(j0 + 64) * ldk, ldk, tid);
            vr.load(V + base + (long)(j0 + 64) * D, D, tid);
        }
        f32x16 st[2];
        mma_rows_x_frags<D>(st, Ks, qf, lr, lh);              // S^T tile: rows = keys, this lane's column = its query
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[t][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));             // the other half-wave holds the other 32 keys
        // (QS: scores, maxima and the share statistics are in binary orders.)  QS keeps the running maximum a query started with unless a
        // tile exceeds it by more than kDeferBits orders (round 6; the ping-pong kernel has done so since round 5): P then lies in
        // (0, 2^kDeferBits] at the same relative bf16 precision, row sum and O scale with it — the same soft-max.  With 16 (level 2) or 4
        // (level 3) key tiles per query SOME lane of the wave met a new maximum in nearly every tile, and the rescale of O — accumulators
        // the compiler keeps in AGPRs: 64 reads + 64 multiplies + 64 writes at D = 128 — ran nearly every tile: 13.7 VALU
        // instructions per MFMA (profiles/r06_attn_l2_sq_pmc.txt).
        const float m_new = QS ? ((mx > m_run + kDeferBits) ? mx : m_run) : fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * (QS ? 1.f : kLog2e));
        const float nm = -m_new * (QS ? 1.f : kLog2e);      // exp(s - m) = 2^(s log2e - m log2e): one fma + v_exp_f32 per score
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(QS ? st[t][r] + nm : fmaf(st[t][r], kLog2e, nm));
                st[t][r] = pv;
                sum += pv;
            }
        l_run = l_run * alpha + sum;
        m_run = m_new;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f)) {     // after the first tiles the running maxima rarely move
#pragma unroll
            for (int ct = 0; ct < D / 32; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;
        }
        mma_tr_x_tile<D>(o, Vs, st, lane);                    // O^T += V^T P^T
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (SPLIT) {
        const long row = ((long)zs * gridDim.y + by) * N + q;
        store_ct<D>(part_o + row * D, o, 1.f, nullptr, lh);
        if (lh == 0) {
            part_ml[2 * row] = m_run;
            part_ml[2 * row + 1] = l_tot;
        }
        return;
    }
    store_ct<D>(out + base + (long)q * D, o, 1.f / l_tot, Vres ? Vres + base + (long)q * D : nullptr, lh);
    if (out16)
        store_ct16<D>(out16 + ((long)by * N + q) * ld16, o, 1.f / l_tot, Vres ? Vres + base + (long)q * D : nullptr, lh);
    if (lh == 0) lse[(long)by * N + q] = QS ? (m_run + __log2f(l_tot)) * kLn2 : m_run + __logf(l_tot);
}

// merge the key shares of the SPLIT forward: out = sum_s w_s O_s / sum_s w_s l_s with w_s = exp(m_s - max_s m_s) (+ V), the bf16
// copy and the log-sum-exp; one thread per (row, four channels)
template <int D, bool QS = false>
__global__ __launch_bounds__(256) void hupr_k_attn_combine(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                          int S, long rows, const float* __restrict__ Vres,
                                                          float* __restrict__ out, float* __restrict__ lse,
                                                          __bf16* __restrict__ out16, int ld16,
                                                          const AttnBatch batch = AttnBatch()) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    if (batch.n > 0) {                                        // blockIdx.y = item
        const int item = blockIdx.y;
        part_o += (long)item * S * rows * D;
        part_ml += (long)item * S * rows * 2;
        Vres = batch.Vres[item];
        out = batch.out[item];
        lse = batch.lse[item];
        out16 = batch.out16[item];
    }
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long row = t / (D / 4);
    const int c = (int)(t % (D / 4)) * 4;
    if (row >= rows) return;
    float m = -INFINITY;
    for (int s = 0; s < S; ++s) m = fmaxf(m, part_ml[2 * (s * rows + row)]);
    float L = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < S; ++s) {
        const float w = __builtin_amdgcn_exp2f((part_ml[2 * (s * rows + row)] - m) * (QS ? 1.f : kLog2e));
        L = fmaf(w, part_ml[2 * (s * rows + row) + 1], L);
        const float4 o = *reinterpret_cast<const float4*>(part_o + (s * rows + row) * D + c);
        acc.x = fmaf(w, o.x, acc.x); acc.y = fmaf(w, o.y, acc.y); acc.z = fmaf(w, o.z, acc.z); acc.w = fmaf(w, o.w, acc.w);
    }
    const float inv = 1.f / L;
    float4 v = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    if (Vres) {
        const float4 a = *reinterpret_cast<const float4*>(Vres + row * D + c);
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
    *reinterpret_cast<float4*>(out + row * D + c) = v;
    if (out16) {
        const bf16x4 o4 = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
        *reinterpret_cast<bf16x4*>(out16 + row * ld16 + c) = o4;
    }
    if (c == 0) lse[row] = QS ? (m + __log2f(L)) * kLn2 : m + __logf(L);
}

// Up to four attentions of one shape in one launch of a backward kernel (n > 0; the four of an MSCSA level).  Row-sum kernel: blockIdx.y
// = item; dQ and dK / dV kernels: grid.y = n * Bn samples, item = sample / Bn.  (dK / dV: the items of ONE launch must write distinct dV.)
struct AttnBwdBatch {
    int n, Bn;
    const void* K[4];
    const void* Q[4];
    const void* V[4];
    const void* dO[4];
    const float* out[4];
    const float* V32[4];
    const float* lse[4];
    float* Dq[4];
    float* dK[4];
    float* dQ[4];
    float* dV[4];
    const float* add32[4];
    const __bf16* add16[4];
    int residual[4];
};

// D[q] = sum_c dO[q,c] * (out[q,c] - (residual ? V[q,c] : 0))
template <int D, typename TG>
__global__ __launch_bounds__(256) void hupr_k_attn_prep(const TG* __restrict__ dO, int lddo, const float* __restrict__ out,
                                                        const float* __restrict__ V, float* __restrict__ Dq, long rows,
                                                        int residual, const AttnBwdBatch batch = AttnBwdBatch()) {
    if (batch.n > 0) {
        const int item = (int)blockIdx.y;
        dO = static_cast<const TG*>(batch.dO[item]);
        out = batch.out[item];
        V = batch.V32[item];
        Dq = batch.Dq[item];
        residual = batch.residual[item];
    }
    // 16 lanes per row (D/16 floats each)
    const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int sub = threadIdx.x & 15;
    float acc = 0.f;
    if (row < rows) {
#pragma unroll
        for (int i = 0; i < D / 64; ++i) {
            const long o = row * D + (sub + 16 * i) * 4;
            const float4 g = ld_act4<TG>(dO + row * lddo + (sub + 16 * i) * 4);
            float4 y = *reinterpret_cast<const float4*>(out + o);
            if (residual) {
                const float4 v = *reinterpret_cast<const float4*>(V + o);
                y.x -= v.x; y.y -= v.y; y.z -= v.z; y.w -= v.w;
            }
            acc += (g.x * y.x + g.y * y.y) + (g.z * y.z + g.w * y.w);
        }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 16);
    if (row < rows && sub == 0) Dq[row] = acc;
}

// ------------------------------------------------------------------------------------------------------
// backward, dQ: same walk as the forward
// ------------------------------------------------------------------------------------------------------
template <int D, typename TI, bool QS = false>
__global__ __launch_bounds__(256, D <= 128 ? 2 : 1) void hupr_k_attn_bwd_dq(const TI* __restrict__ K, const TI* __restrict__ Q,
                                                          const TI* __restrict__ V, const TI* __restrict__ dO,
                                                          const float* __restrict__ lse, const float* __restrict__ Dq,
                                                          float* __restrict__ dQ, int N, int ldk, int ldq, int lddq, int lddo, int xcd_map,
                                                          const AttnBwdBatch batch = AttnBwdBatch()) {
    // D = 128 (level 2): key tiles of 32 since round 6 — two S^T / dP^T accumulator tiles instead of four, half the staging registers:
    // <= 256 unified registers = two workgroups per CU (358 = one wave per SIMD before: 156 us for the level's four attentions); the same
    // MFMA sequence per query (the 16-key blocks arrive in the same order): the same bits
    // (The dK / dV kernel at D = 128 — 446 registers — does not follow: with 32-query tiles, the V rows in LDS and the dO rows staged
    // synchronously it still spills 19-41 registers at 256 and ran 131 us instead of 116, profiles/r06_attn_dq128_ab.txt.)
    constexpr int KT = D == 128 ? 32 : 64, NT = KT / 32;      // (D = 64 on 32-key tiles — 156 registers, three workgroups per CU: 789 vs 781 us, no gain)
    __shared__ __attribute__((aligned(16))) __bf16 Ks[KT * D];
    __shared__ __attribute__((aligned(16))) __bf16 Vs[KT * D];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
    int bx, by;
    xcd_block(bx, by, xcd_map);
    if (batch.n > 0) {
        const int item = by / batch.Bn;
        by -= item * batch.Bn;
        K = static_cast<const TI*>(batch.K[item]);
        Q = static_cast<const TI*>(batch.Q[item]);
        V = static_cast<const TI*>(batch.V[item]);
        dO = static_cast<const TI*>(batch.dO[item]);
        lse = batch.lse[item];
        Dq = batch.Dq[item];
        dQ = batch.dQ[item];
    }
    const long base = (long)by * N * D;
    const int q = bx * 128 + wave * 32 + lr;
    K += (long)by * N * ldk;
    bf16x8 qf[D / 16], gf[D / 16];
    load_frags<D, TI>(qf, Q + ((long)by * N + q) * ldq, lh);
    load_frags<D, TI>(gf, dO + ((long)by * N + q) * lddo, lh);
    const float nlse_q = -lse[(long)by * N + q] * kLog2e, d_q = Dq[(long)by * N + q];
    f32x16 nlse16;                                            // (QS) accumulator input of every S^T chain of this lane's query
#pragma unroll
    for (int r = 0; r < 16; ++r) nlse16[r] = nlse_q;
    f32x16 dq[D / 32];
#pragma unroll
    for (int ct = 0; ct < D / 32; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[ct][r] = 0.f;
    constexpr bool PF = (D <= 128);
    StageRegs<PF ? D : 64, KT, TI> kr, vr;
    if (PF) {
        kr.load(K, ldk, tid);
        vr.load(V + base, D, tid);
    }
    for (int j0 = 0; j0 < N; j0 += KT) {
        __syncthreads();
        if (PF) {
            kr.store(Ks, tid);
            vr.store(Vs, tid);
        } else {
            stage_rows<D, KT, TI>(Ks, K + (long)j0 * ldk, ldk, tid);
            stage_rows<D, KT, TI>(Vs, V + base + (long)j0 * D, D, tid);
        }
        __syncthreads();
        if (PF && j0 + KT < N) {
            kr.load(K + (long)(j0 + KT) * ldk, ldk, tid);
            vr.load(V + base + (long)(j0 + KT) * D, D, tid);
        }
        f32x16 st[NT], dp[NT];
        if constexpr (QS) {                                   // the chain starts from -log-sum-exp (binary orders): S^T leaves as the exponent
#pragma unroll
            for (int t = 0; t < NT; ++t) st[t] = nlse16;
            mma_rows_x_frags<D, false, NT>(st, Ks, qf, lr, lh);
        } else {
            mma_rows_x_frags<D, true, NT>(st, Ks, qf, lr, lh);          // S^T
        }
        mma_rows_x_frags<D, true, NT>(dp, Vs, gf, lr, lh);    // dP^T = V dO^T
        {      // dS^T on accumulator pairs (packed fma / add / mul: same roundings, half the VALU instructions)
            typedef float f32x2a __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2a arg = QS ? (f32x2a){st[t][r], st[t][r + 1]}
                                          : __builtin_elementwise_fma((f32x2a){st[t][r], st[t][r + 1]}, (f32x2a){kLog2e, kLog2e}, (f32x2a){nlse_q, nlse_q});
                    const f32x2a pv = {__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
                    const f32x2a ds = pv * ((f32x2a){dp[t][r], dp[t][r + 1]} - (f32x2a){d_q, d_q});
                    st[t][r] = ds[0];
                    st[t][r + 1] = ds[1];
                }
        }
        mma_tr_x_tile<D, D / 32, NT>(dq, Ks, st, lane);       // dQ^T += K^T dS^T
    }
    store_ct<D>(dQ + ((long)by * N + q) * lddq, dq, 1.f, nullptr, lh);
}

// ------------------------------------------------------------------------------------------------------
// backward, dK / dV: a workgroup owns 128 keys (32 per wave) and streams 64-query tiles
// ------------------------------------------------------------------------------------------------------
template <int D, typename TI, int NH, bool QS = false>
__global__ __launch_bounds__(256, D == 64 ? 2 : 1) void hupr_k_attn_bwd_dkv(const TI* __restrict__ K, const TI* __restrict__ Q,
                                                           const TI* __restrict__ V, const TI* __restrict__ dO,
                                                           const float* dVadd,
                                                           const float* __restrict__ lse, const float* __restrict__ Dq,
                                                           float* __restrict__ dK, float* dV, int N, int ldk,
                                                           int ldq, int lddk, int lddo, const __bf16* dVadd16, int ldadd16, int xcd_map,
                                                           const AttnBwdBatch batch = AttnBwdBatch()) {
    __shared__ __attribute__((aligned(16))) __bf16 Qs[64 * D];
    __shared__ __attribute__((aligned(16))) __bf16 Gs[64 * D];
    __shared__ __attribute__((aligned(16))) float s_lse[64], s_d[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
    int bx, by;
    xcd_block(bx, by, xcd_map);
    if (batch.n > 0) {
        const int item = by / batch.Bn;
        by -= item * batch.Bn;
        K = static_cast<const TI*>(batch.K[item]);
        Q = static_cast<const TI*>(batch.Q[item]);
        V = static_cast<const TI*>(batch.V[item]);
        dO = static_cast<const TI*>(batch.dO[item]);
        lse = batch.lse[item];
        Dq = batch.Dq[item];
        dK = batch.dK[item];
        dV = batch.dV[item];
        dVadd = batch.add32[item];
        dVadd16 = batch.add16[item];
    }
    const long base = (long)by * N * D;
    const int key = bx * 128 + wave * 32 + lr;                // this lane's key
    bf16x8 kf[D / 16], vf[D / 16];
    load_frags<D, TI>(kf, K + ((long)by * N + key) * ldk, lh);
    load_frags<D, TI>(vf, V + base + (long)key * D, lh);
    Q += (long)by * N * ldq;
    dO += (long)by * N * lddo;
    // NH = 2 (D = 256): blockIdx.z picks the half of the OUTPUT channels this launch slice accumulates (S, dS are
    // recomputed per half — the full set of dK, dV accumulators would not fit the register file)
    constexpr int DV = D / NH;
    const int c0 = blockIdx.z * DV;
    f32x16 dk[DV / 32], dv[DV / 32];
#pragma unroll
    for (int ct = 0; ct < DV / 32; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[ct][r] = 0.f; dv[ct][r] = 0.f; }
    // register prefetch of the next tile: both arrays at D = 128 (one wave per SIMD, 512 registers); at D = 64 the
    // kernel sits at the 256-register limit of two waves per SIMD and only the query rows fit
    constexpr bool PFQ = (D <= 128), PFG = (D == 128);
    StageRegs<PFQ ? D : 64, 64, TI> qr;
    StageRegs<PFG ? D : 64, 64, TI> gr;
    if (PFQ) qr.load(Q, ldq, tid);
    if (PFG) gr.load(dO, lddo, tid);
    for (int q0 = 0; q0 < N; q0 += 64) {
        __syncthreads();
        if (PFQ) qr.store(Qs, tid);
        else stage_rows<D, 64, TI>(Qs, Q + (long)q0 * ldq, ldq, tid);
        if (PFG) gr.store(Gs, tid);
        else stage_rows<D, 64, TI>(Gs, dO + (long)q0 * lddo, lddo, tid);
        if (tid < 64) {
            s_lse[tid] = -lse[(long)by * N + q0 + tid] * kLog2e;              // pre-scaled for the exp2 form
            s_d[tid] = Dq[(long)by * N + q0 + tid];
        }
        __syncthreads();
        if (q0 + 64 < N) {
            if (PFQ) qr.load(Q + (long)(q0 + 64) * ldq, ldq, tid);
            if (PFG) gr.load(dO + (long)(q0 + 64) * lddo, lddo, tid);
        }
        f32x16 s[2], dp[2];
        if constexpr (QS) {                                   // rows = queries: register r of tile t starts from -log-sum-exp of ITS query
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float4 l4 = *reinterpret_cast<const float4*>(&s_lse[32 * t + 8 * r4 + 4 * lh]);
                    s[t][4 * r4] = l4.x; s[t][4 * r4 + 1] = l4.y; s[t][4 * r4 + 2] = l4.z; s[t][4 * r4 + 3] = l4.w;
                }
            mma_rows_x_frags<D, false>(s, Qs, kf, lr, lh);
        } else {
            mma_rows_x_frags<D>(s, Qs, kf, lr, lh);           // S tile: rows = queries, this lane's column = its key
        }
        mma_rows_x_frags<D>(dp, Gs, vf, lr, lh);              // dP = dO V^T
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qi = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float pv = __builtin_amdgcn_exp2f(QS ? s[t][r] : fmaf(s[t][r], kLog2e, s_lse[qi]));
                dp[t][r] = pv * (dp[t][r] - s_d[qi]);          // dS
                s[t][r] = pv;                                   // P
            }
        mma_tr_x_tile<D, DV / 32>(dv, Gs, s, lane, c0 / 32);  // dV^T += dO^T P
        mma_tr_x_tile<D, DV / 32>(dk, Qs, dp, lane, c0 / 32); // dK^T += Q^T dS
    }
    store_ct<DV>(dK + ((long)by * N + key) * lddk + c0, dk, QS ? kLn2 : 1.f, nullptr, lh);      // (QS: dK^T was summed over Q' = log2e Q)
    if (dVadd16) store_ct_add16<DV>(dV + base + (long)key * D + c0, dv, dVadd16 + ((long)by * N + key) * ldadd16 + c0, lh);
    else store_ct<DV>(dV + base + (long)key * D + c0, dv, 1.f, dVadd ? dVadd + base + (long)key * D + c0 : nullptr, lh);
}


// dK / dV at the level-1 shape (D = 64, bf16 operands): 512 threads own 256 keys (32 per wave) and stream the 64-query tiles through
// DOUBLE-buffered images with ONE barrier per tile, both tiles (Q and dO) prefetched into registers one tile ahead.  What the SQ
// counters said about the 256-thread kernel above at this shape (profiles/r04b_attn_sq_pmc.txt): the waves parked 38 % of their
// cycles — it sits at the register limit of two waves per SIMD, so only the query rows were prefetched and every tile waited for the
// global loads of its dO rows between two barriers.  With eight waves sharing a tile a thread stages 16 bytes of each image instead
// of 32 of one, and both prefetches fit in the registers one used.  Same arithmetic per (key, query tile), same tile order: the
// same bits as the kernel above.
template <bool QS>
__global__ __launch_bounds__(512) void hupr_k_attn_bwd_dkv512(const __bf16* __restrict__ K, const __bf16* __restrict__ Q,
                                                              const __bf16* __restrict__ V, const __bf16* __restrict__ dO,
                                                              const float* dVadd, const float* __restrict__ lse,
                                                              const float* __restrict__ Dq, float* __restrict__ dK, float* dV, int N,
                                                              int ldk, int ldq, int lddk, int lddo, const __bf16* dVadd16,
                                                              int ldadd16, int xcd_map) {
    constexpr int D = 64;
    __shared__ __attribute__((aligned(16))) __bf16 Qs[2][64 * D];
    __shared__ __attribute__((aligned(16))) __bf16 Gs[2][64 * D];
    __shared__ __attribute__((aligned(16))) float s_lse[2][64], s_d[2][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
    int bx, by;
    xcd_block(bx, by, xcd_map);
    const long base = (long)by * N * D;
    const int key = bx * 256 + wave * 32 + lr;                // this lane's key
    bf16x8 kf[D / 16], vf[D / 16];
    load_frags<D, __bf16>(kf, K + ((long)by * N + key) * ldk, lh);
    load_frags<D, __bf16>(vf, V + base + (long)key * D, lh);
    Q += (long)by * N * ldq;
    dO += (long)by * N * lddo;
    f32x16 dk[D / 32], dv[D / 32];
#pragma unroll
    for (int ct = 0; ct < D / 32; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[ct][r] = 0.f; dv[ct][r] = 0.f; }
    // staging: thread t moves chunk t & 7 (16 bytes) of row t >> 3 of both images; threads 0..63 / 64..127 the row statistics
    const int srow = tid >> 3, sch = tid & 7;
    const __bf16* qp = Q + (long)srow * ldq + sch * 8;
    const __bf16* gp = dO + (long)srow * lddo + sch * 8;
    const float* sp = tid < 64 ? lse + (long)by * N + tid : Dq + (long)by * N + (tid & 63);
    u32x4a qreg = *reinterpret_cast<const u32x4a*>(qp), greg = *reinterpret_cast<const u32x4a*>(gp);
    float sreg = tid < 128 ? *sp : 0.f;
    *reinterpret_cast<u32x4a*>(&Qs[0][Img<D>::off(srow, sch)]) = qreg;
    *reinterpret_cast<u32x4a*>(&Gs[0][Img<D>::off(srow, sch)]) = greg;
    if (tid < 64) s_lse[0][tid] = -sreg * kLog2e;             // pre-scaled for the exp2 form
    else if (tid < 128) s_d[0][tid - 64] = sreg;
    __syncthreads();
    for (int q0 = 0; q0 < N; q0 += 64) {
        const int cb = (q0 >> 6) & 1;
        if (q0 + 64 < N) {                                     // the next tile travels while this one is multiplied
            qreg = *reinterpret_cast<const u32x4a*>(qp + (long)(q0 + 64) * ldq);
            greg = *reinterpret_cast<const u32x4a*>(gp + (long)(q0 + 64) * lddo);
            if (tid < 128) sreg = sp[q0 + 64];
        }
        f32x16 s[2], dp[2];
        if constexpr (QS) {                                    // register r of tile t starts from -log-sum-exp of ITS query (binary orders)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float4 l4 = *reinterpret_cast<const float4*>(&s_lse[cb][32 * t + 8 * r4 + 4 * lh]);
                    s[t][4 * r4] = l4.x; s[t][4 * r4 + 1] = l4.y; s[t][4 * r4 + 2] = l4.z; s[t][4 * r4 + 3] = l4.w;
                }
            mma_rows_x_frags<D, false>(s, Qs[cb], kf, lr, lh);
        } else {
            mma_rows_x_frags<D>(s, Qs[cb], kf, lr, lh);        // S tile: rows = queries, this lane's column = its key
        }
        mma_rows_x_frags<D>(dp, Gs[cb], vf, lr, lh);           // dP = dO V^T
        // P and dS on accumulator PAIRS (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32: the same roundings in half the VALU
        // instructions — 160 VALU per tile beside 32 MFMAs were as many issue cycles as the matrix pipe's own)
        typedef float f32x2a __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int qi = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const f32x2a dd = *reinterpret_cast<const f32x2a*>(&s_d[cb][qi]);
                f32x2a arg = {s[t][r], s[t][r + 1]};
                if constexpr (!QS) arg = __builtin_elementwise_fma(arg, (f32x2a){kLog2e, kLog2e}, *reinterpret_cast<const f32x2a*>(&s_lse[cb][qi]));
                const f32x2a pv = {__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
                const f32x2a ds = pv * ((f32x2a){dp[t][r], dp[t][r + 1]} - dd);
                dp[t][r] = ds[0];                               // dS
                dp[t][r + 1] = ds[1];
                s[t][r] = pv[0];                                // P
                s[t][r + 1] = pv[1];
            }
        mma_tr_x_tile<D, D / 32>(dv, Gs[cb], s, lane, 0);      // dV^T += dO^T P
        mma_tr_x_tile<D, D / 32>(dk, Qs[cb], dp, lane, 0);     // dK^T += Q^T dS
        if (q0 + 64 < N) {                                     // the other buffers were last read one tile ago, behind a barrier
            *reinterpret_cast<u32x4a*>(&Qs[cb ^ 1][Img<D>::off(srow, sch)]) = qreg;
            *reinterpret_cast<u32x4a*>(&Gs[cb ^ 1][Img<D>::off(srow, sch)]) = greg;
            if (tid < 64) s_lse[cb ^ 1][tid] = -sreg * kLog2e;
            else if (tid < 128) s_d[cb ^ 1][tid - 64] = sreg;
        }
        __syncthreads();
    }
    store_ct<D>(dK + ((long)by * N + key) * lddk, dk, QS ? kLn2 : 1.f, nullptr, lh);      // (QS: dK^T was summed over Q' = log2e Q)
    if (dVadd16) store_ct_add16<D>(dV + base + (long)key * D, dv, dVadd16 + ((long)by * N + key) * ldadd16, lh);
    else store_ct<D>(dV + base + (long)key * D, dv, 1.f, dVadd ? dVadd + base + (long)key * D : nullptr, lh);
}

// ------------------------------------------------------------------------------------------------------
// Ping-pong kernels (round 4; D = 64, bf16 operands) — the MSCSA level-1 shape (N = 4096), 88 % of the attention flops
// ------------------------------------------------------------------------------------------------------
// What the measurements of rounds 2-3 said about the kernels above at D = 64: per 64-key tile a wave
// issues 16 MFMAs (512 matrix-pipe cycles) and ~176 VALU instructions (~840 cycles); a second workgroup on the CU adds its own
// 1 800 cycles per tile instead of hiding in the first one's stalls — the two waves of a SIMD run the same phases in step
// (matrix beside matrix, soft-max beside soft-max), and a kernel without MFMAs and exponentials still takes 153 of 206 us:
// register-staged K / V tiles (global -> VGPR -> ds_write, two barriers per tile) and fragment reads in front of their MFMAs.
// Here a 512-thread workgroup owns 256 queries (32 per wave; two waves per SIMD) and every wave software-pipelines a key tile
// INSIDE itself: the soft-max of tile j is interleaved, instruction group by instruction group, with the MFMAs of its
// neighbours that do not depend on it (O^T += V^T P^T of tile j - 1 under the row maxima, S^T of tile j + 1 under the
// exponentials), with ONE barrier per key tile.  Two earlier builds held the two wave groups (waves 0-3 / 4-7) in anti-phase
// instead — first with two barriers per tile (strict alternation), then with one barrier taken at different points — so that
// one wave of a SIMD multiplies while its partner exponentiates: the soft-max segments (1 300-1 650 cycles against 600 for the
// MFMAs) then run one after the other; all three builds are within 3 % (stamps in profiles/r04_attn_trace.txt, ablations
// in profiles/r04_attn_ablation.txt; DESIGN.md section 7 has the clock finding that explains why the phases add).
//   * K / V tiles travel global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, the swizzle applied on the source side) through
//     a ring of eight 16 KB slots, four tiles ahead: no staging registers, no ds_write, no barrier of their own (each wave
//     covers its own pieces with a counted vmcnt before the tile barrier; a slot is rewritten five tiles after its last read);
//   * the soft-max works on score PAIRS (v_pk_fma_f32 / v_pk_add_f32, v_max3_f32) and exchanges the half-waves' maxima with one
//     v_permlane32_swap instead of an LDS shuffle: ~105 VALU instructions per tile;
//   * the workgroups of one sample share an XCD (one L2 fetches its K / V once: 1 MB per sample instead of up to 8);
//   * same tile order, same running maximum and the same bf16 rounding of P as the kernels above: results equal up to the
//     association of the row sums.
typedef float v2fa __attribute__((ext_vector_type(2)));
constexpr int kPPRing = 8, kPPAhead = 4;

__device__ __forceinline__ void pp_dma(unsigned m0val, int voff, u32x4a rsrc, int soff) {
    unsigned keep;
    m0val = __builtin_amdgcn_readfirstlane(m0val);          // wave-uniform by construction; the "s" constraint does not enforce it
    soff = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(m0val), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}
// all but the youngest N_ vector-memory operations of this wave have completed (vmcnt is 6 bits: [3:0] and [15:14])
#define HUPR_PP_VMCNT(N_) __builtin_amdgcn_s_waitcnt(0x0F70 | ((N_) & 15) | (((N_) >> 4) << 14))
#define HUPR_PP_BARRIER()                    \
    __builtin_amdgcn_sched_barrier(0);       \
    __builtin_amdgcn_s_barrier();            \
    __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ u32x4a pp_rsrc(const void* base, long bytes) {
    return (u32x4a){(unsigned)(unsigned long)base, (unsigned)((unsigned long)base >> 32) & 0xffffu, (unsigned)bytes, 0x00020000u};
}

// lane -> (sample, query block) of a 1-D grid: workgroup id L runs on XCD L % 8; the query blocks of one sample stay on one XCD
__device__ __forceinline__ void pp_block(int nqb, int Bn, int& b, int& qb) {
    const int L = blockIdx.x;
    if (Bn % 8 == 0) {
        const int idx = L >> 3;
        qb = idx % nqb;
        b = (L & 7) + 8 * (idx / nqb);
    } else {
        qb = L % nqb;
        b = L / nqb;
    }
}

template <bool QS = false>
__global__ __launch_bounds__(512, 1) void hupr_k_attn_fwd_pp64(const __bf16* __restrict__ K, const __bf16* __restrict__ Q,
                                                              const __bf16* __restrict__ V, const float* __restrict__ Vres,
                                                              float* __restrict__ out, float* __restrict__ lse, int N, int Bn, int ldk,
                                                              int ldq, __bf16* __restrict__ out16, int ld16,
                                                              unsigned long long* __restrict__ trace) {
    // (phase ablations of this kernel — no LDS-DMA / fragment reads / exponentials / maxima / MFMAs / barrier / stores — were timed in
    // round 4: profiles/r04_attn_ablation.txt; the instantiations are gone)
    constexpr int D = 64;
    __shared__ __attribute__((aligned(1024))) __bf16 ring[kPPRing][2][64 * D];      // [slot][K image, V image]
    // profiling (scripts/attn_pp_trace.py): s_memtime stamps of waves 0 and 4 of workgroup 0, five per key tile
    const bool tracing = trace != nullptr && blockIdx.x == 0 && (threadIdx.x >> 6) % 4 == 0;
    unsigned long long* tr = trace + (threadIdx.x >> 8) * 4096;
    int tslot = 0;
#define HUPR_PP_STAMP()                                                             \
    if (tracing) {                                                                  \
        const unsigned long long t__ = __builtin_amdgcn_s_memtime();                \
        if ((threadIdx.x & 63) == 0 && tslot < 4096) tr[tslot] = t__;               \
        ++tslot;                                                                    \
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
    const int grp = wave >> 2;
    // static issue priority for one of the two waves of each SIMD.  Stamps of the
    // equal-priority build: wave 0's tile 1 820 cycles + 860 at the barrier, wave 4's 2 470 + 220; measured 171.1 (waves 4-7) /
    // 171.5 (none) / 167.4 us (waves 0-3)
    if (grp == 0) __builtin_amdgcn_s_setprio(1);
    int b, qb;
    pp_block(N / 256, Bn, b, qb);
    const long base = (long)b * N * D;
    const int q = qb * 256 + wave * 32 + lr;                  // this lane's query
    const int nt = N / 64;
    bf16x8 qf[4];
    load_frags<D, __bf16>(qf, Q + ((long)b * N + q) * ldq, lh);

    // LDS-DMA: wave w deposits rows 8 w .. 8 w + 7 of the K image and of the V image of a tile (1 KiB each); lane l fills row
    // 8 w + (l >> 3), chunk position l & 7, with source chunk (l & 7) ^ key(row)
    const int prow = 8 * wave + (lane >> 3);
    const int pchunk = (lane & 7) ^ Img<D>::key(prow);
    const int kvoff = (prow * ldk + pchunk * 8) * 2, vvoff = (prow * D + pchunk * 8) * 2;
    const u32x4a krs = pp_rsrc(K + (long)b * N * ldk, (long)N * ldk * 2), vrs = pp_rsrc(V + base, (long)N * D * 2);
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)&ring[0][0][0];
    const unsigned piece = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024);
#define HUPR_PP_ISSUE(TILE_)                                                                    \
    {                                                                                           \
        const int tl_ = min((TILE_), nt - 1), sl_ = (TILE_) & (kPPRing - 1);                    \
        pp_dma(piece + sl_ * 16384, kvoff, krs, tl_ * 64 * ldk * 2);                            \
        pp_dma(piece + sl_ * 16384 + 8192, vvoff, vrs, tl_ * 64 * D * 2);                       \
    }
    // per-lane fragment addresses inside a slot (bf16 elements)
    int koff[4];                                              // K rows 32 t + lr, K-step ks: + t * 32 * D
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = Img<D>::off(lr, 2 * ks + lh);
    const int g = lane >> 4, s16 = lane & 15, h = g >> 1;
    const int col = 16 * (g & 1) + 4 * (s16 & 3), rsub = s16 >> 2;
    int voff0[2], voff1[2];                                   // V rows row0 = 4 h + rsub (+ 32 t + 16 u), row1 = row0 + 8; channel tile ct
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int c = 32 * ct + col;
        voff0[ct] = Img<D>::off(4 * h + rsub, c >> 3) + (c & 4);
        voff1[ct] = Img<D>::off(4 * h + rsub + 8, c >> 3) + (c & 4);
    }

    f32x16 o[2], st[2][2];                                    // st[parity]: S^T of the current tile / of the next one
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] = 0.f;
    // QS: m_run is the running maximum in binary orders as DEFERRED (see kDeferBits), starting from 0 — the first tile always takes
    // the rescale branch — and negm16 (every element -m_run) is the accumulator input of the next tile's S^T chain
    float m_run = QS ? 0.f : -INFINITY, l_run = 0.f;
    f32x16 negm16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bf16x8 kf[2][4], vf[2][2][2], pb[2][2][2];                // pb[parity]: P^T of the current tile / of the previous one

#define HUPR_PP_READ_K(SLOT_)                                                                                      \
    {                                                                                                              \
        const __bf16* ki_ = &ring[SLOT_][0][0];                                                                    \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                           \
            _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                          \
                kf[t][ks] = *reinterpret_cast<const bf16x8*>(ki_ + koff[ks] + t * 32 * D);                         \
    }
#define HUPR_PP_READ_V(SLOT_)                                                                                      \
    {                                                                                                              \
        const __bf16* vi_ = &ring[SLOT_][1][0];                                                                    \
        _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                              \
            _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                          \
                _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) {                                                 \
                    const s16x4 lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16(                                     \
                        (__attribute__((address_space(3))) s16x4*)(vi_ + voff0[ct] + (32 * t + 16 * u) * D));     \
                    const s16x4 hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16(                                     \
                        (__attribute__((address_space(3))) s16x4*)(vi_ + voff1[ct] + (32 * t + 16 * u) * D));     \
                    union { struct { s16x4 a, b; } s2; bf16x8 v; } uu_;                                            \
                    uu_.s2.a = lo_;                                                                                \
                    uu_.s2.b = hi_;                                                                                \
                    vf[t][u][ct] = uu_.v;                                                                          \
                }                                                                                                  \
    }
#define HUPR_PP_S(DST_)                                                                                            \
    {                                                                                                              \
        _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                              \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) st[DST_][t][r] = 0.f;                                   \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                           \
            _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                          \
                st[DST_][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[t][ks], qf[ks], st[DST_][t], 0, 0, 0);    \
    }
#define HUPR_PP_PV(SRC_)                                                                                           \
    _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                                  \
        _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                              \
            _Pragma("unroll") for (int ct = 0; ct < 2; ++ct)                                                       \
                o[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[t][u][ct], pb[SRC_][t][u], o[ct], 0, 0, 0);

    // prologue: kPPAhead tiles in flight, the first two landed and published
    HUPR_PP_ISSUE(0)
    HUPR_PP_ISSUE(1)
    HUPR_PP_ISSUE(2)
    HUPR_PP_ISSUE(3)
    HUPR_PP_VMCNT(4);
    HUPR_PP_BARRIER();
    HUPR_PP_READ_K(0)
    HUPR_PP_S(0)

    // One key tile j, software-pipelined INSIDE the wave: the soft-max of tile j (VALU: st[CUR_] -> pb[CUR_]) is interleaved,
    // instruction group by instruction group, with the MFMAs of its neighbours, which do not depend on it — O^T += V^T P^T of
    // tile j - 1 (pb[NXT_], V fragments read at the end of the previous iteration) under the row maxima, S^T of tile j + 1
    // (-> st[NXT_]) under the exponentials.  A wave issues a VALU instruction every ~5 cycles at best (8.4 when dependent,
    // v_exp_f32 9-12: scripts/probes/valu_issue_probe.hip) while a 32x32x16 MFMA occupies the matrix pipe for 32 cycles: five to
    // six VALU instructions fit in the shadow of each MFMA of the SAME wave, whereas two waves that alternate whole phases each
    // wait out their own issue latencies (the first builds of this kernel: soft-max segment 1 300-1 500 cycles, matrix segment
    // 600, a tile 2 900).  hipcc does not build such a schedule from sched_group_barrier patterns here (it clumped the sixteen
    // MFMAs behind the maxima, each behind its own lgkmcnt(0)), so the order is written out: one MFMA + its share of the VALU
    // work per group, groups fenced with sched_barrier(0).  FIRST_ / LAST_ (compile-time): no previous tile / no next tile.
#define HUPR_SB() __builtin_amdgcn_sched_barrier(0)
#define HUPR_PVM(T_, U_, CT_)                                                                                              \
    if (!(FIRST__)) o[CT_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[T_][U_][CT_], pb[NXT__][T_][U_], o[CT_], 0, 0, 0);
    // The MFMA of a group is pinned between two empty asm statements through its A operand: the first one "defines" the fragment
    // (after the previous group's statement), the second one "redefines" it, so the MFMA that reads it sits in between — without
    // the statements the instruction selector linearises all eight MFMAs behind the exponentials (their results are needed last).
#define HUPR_SM(KS_, T_)                                                                                                   \
    if (!(LAST__)) {                                                                                        \
        asm volatile("" : "+v"(kf[T_][KS_]));                                                                              \
        if ((KS_) == 0) st[NXT__][T_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[T_][0], qf[0], QS ? negm16 : zero16, 0, 0, 0); \
        else st[NXT__][T_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[T_][KS_], qf[KS_], st[NXT__][T_], 0, 0, 0);        \
        asm volatile("" : "+v"(kf[T_][KS_]));                                                                              \
    }
#define HUPR_M3(A_, B_, C_) fmaxf(fmaxf(A_, B_), C_)
    // scores 2 P_, 2 P_ + 1 of the current tile (P_ = 0..15: t = P_ >> 3, u = (P_ >> 2) & 1, i = 2 (P_ & 3))
#define HUPR_PAIR(P_)                                                                                                      \
    {                                                                                                                      \
        constexpr int t_ = (P_) >> 3, u_ = ((P_) >> 2) & 1, i_ = 2 * ((P_) & 3);                                           \
        v2fa s2_ = {st[CUR__][t_][8 * u_ + i_], st[CUR__][t_][8 * u_ + i_ + 1]};                                           \
        asm volatile("" : "+v"(s2_));                                                                                      \
        const v2fa e2_ = QS ? s2_ : __builtin_elementwise_fma(s2_, l2e2, nm2);                                             \
        const float p0_ = __builtin_amdgcn_exp2f(e2_.x);                                                         \
        const float p1_ = __builtin_amdgcn_exp2f(e2_.y);                                                         \
        sum2[(P_) & 1] += (v2fa){p0_, p1_};                                                                                \
        pb[CUR__][t_][u_][i_] = (__bf16)p0_;                                                                               \
        pb[CUR__][t_][u_][i_ + 1] = (__bf16)p1_;                                                                           \
        if (((P_) & 1) == 1) asm volatile("" : "+v"(sum2[0]), "+v"(sum2[1]));                                              \
    }
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define HUPR_PP_TILE(CUR_, NXT_, FIRST_, LAST_)                                                                            \
    {                                                                                                                      \
        constexpr int CUR__ = CUR_, NXT__ = NXT_;                                                                          \
        constexpr bool FIRST__ = FIRST_, LAST__ = LAST_;                                                                   \
        HUPR_PP_STAMP()                                                                                                    \
        { HUPR_PP_ISSUE(j + kPPAhead) }                                                                           \
        if (!LAST__) { HUPR_PP_READ_K             ((j + 1) & (kPPRing - 1)) }                                             \
        HUPR_SB();                                                                                                         \
        /* ---- row maxima of tile j under O^T += V^T P^T of tile j - 1 (four independent chains) ---- */                  \
        const f32x16& s0_ = st[CUR__][0];                                                                                  \
        const f32x16& s1_ = st[CUR__][1];                                                                                  \
        float ma_, mb_, mc_, md_;                                                                                          \
        HUPR_PVM(0, 0, 0) ma_ = HUPR_M3(s0_[0], s0_[1], s0_[2]); mb_ = HUPR_M3(s0_[8], s0_[9], s0_[10]); HUPR_SB();        \
        HUPR_PVM(0, 0, 1) mc_ = HUPR_M3(s1_[0], s1_[1], s1_[2]); md_ = HUPR_M3(s1_[8], s1_[9], s1_[10]); HUPR_SB();        \
        HUPR_PVM(0, 1, 0) ma_ = HUPR_M3(ma_, s0_[3], s0_[4]); mb_ = HUPR_M3(mb_, s0_[11], s0_[12]); HUPR_SB();             \
        HUPR_PVM(0, 1, 1) mc_ = HUPR_M3(mc_, s1_[3], s1_[4]); md_ = HUPR_M3(md_, s1_[11], s1_[12]); HUPR_SB();             \
        HUPR_PVM(1, 0, 0) ma_ = HUPR_M3(ma_, s0_[5], s0_[6]); mb_ = HUPR_M3(mb_, s0_[13], s0_[14]); HUPR_SB();             \
        HUPR_PVM(1, 0, 1) mc_ = HUPR_M3(mc_, s1_[5], s1_[6]); md_ = HUPR_M3(md_, s1_[13], s1_[14]); HUPR_SB();             \
        HUPR_PVM(1, 1, 0) ma_ = HUPR_M3(ma_, s0_[7], mb_); mc_ = HUPR_M3(mc_, s1_[7], md_); HUPR_SB();                     \
        HUPR_PVM(1, 1, 1) ma_ = HUPR_M3(ma_, s0_[15], s1_[15]);                                                            \
        float mx = fmaxf(ma_, mc_);                                                                                     \
        {                                                                                                                \
            /* the other half-wave holds the other 32 keys: v_permlane32_swap leaves the lower half-wave's maximum in every  */ \
            /* lane of its first register and the upper one's in the second.  Written as asm: through the builtin hipcc       */ \
            /* dropped the second result (max(r0, r1) compiled to r0 alone, the upper half-wave's maximum was lost — results  */ \
            /* stayed normalised, P was rounded relative to the wrong maximum; found with scripts/attn_pp_ab.py)             */ \
            unsigned ua_ = __builtin_bit_cast(unsigned, mx), ub_ = ua_;                                                    \
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(ua_), "+v"(ub_));                       \
            mx = fmaxf(__builtin_bit_cast(float, ua_), __builtin_bit_cast(float, ub_));                                    \
        }                                                                                                                  \
        float m_new = fmaxf(m_run, mx);                                                                                    \
        float alpha = 1.f, nm = 0.f;                                                                                       \
        if constexpr (QS) {                                                                                                \
            /* the scores of this tile left the matrix pipe relative to m_run (binary orders); keep it unless the tile      */ \
            /* exceeds it by more than kDeferBits — then (rare; always in the first tile, whose chain started from 0) move  */ \
            /* the scores, O (complete: the last MFMA of tile j - 1 was issued above), the row sum and the next chain's     */ \
            /* accumulator input to the new maximum                                                                        */ \
            const bool resc_ = FIRST__ || (mx > kDeferBits);                                                               \
            m_new = m_run;                                                                                                 \
            if (FIRST__ || __builtin_amdgcn_ballot_w64(resc_)) {                                                           \
                const float delta_ = resc_ ? mx : 0.f;                                                                     \
                _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                              \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) st[CUR__][t][r] -= delta_;                              \
                if (!FIRST__) {                                                                                            \
                    const float a_ = __builtin_amdgcn_exp2f(-delta_);                                                      \
                    _Pragma("unroll") for (int ct = 0; ct < 2; ++ct)                                                       \
                        _Pragma("unroll") for (int r = 0; r < 16; ++r) o[ct][r] *= a_;                                     \
                    l_run *= a_;                                                                                           \
                }                                                                                                          \
                m_new = m_run + delta_;                                                                                    \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) negm16[r] = -m_new;                                         \
            }                                                                                                              \
        } else {                                                                                                           \
            alpha = __builtin_amdgcn_exp2f((m_run - m_new) * kLog2e);                                                      \
            nm = -m_new * kLog2e;                             /* exp(s - m) = 2^(s log2e - m log2e) */                     \
        }                                                                                                                  \
        const v2fa nm2 = {nm, nm}, l2e2 = {kLog2e, kLog2e};                                                                \
        v2fa sum2[2] = {{0.f, 0.f}, {0.f, 0.f}};                                                                           \
        HUPR_SB();                                                                                                         \
        /* ---- exponentials of tile j under S^T of tile j + 1: one MFMA + two score pairs per group ---- */               \
        HUPR_SM(0, 0) HUPR_PAIR(0) HUPR_PAIR(1) HUPR_SB();                                                                 \
        HUPR_SM(0, 1) HUPR_PAIR(2) HUPR_PAIR(3) HUPR_SB();                                                                 \
        HUPR_SM(1, 0) HUPR_PAIR(4) HUPR_PAIR(5) HUPR_SB();                                                                 \
        HUPR_SM(1, 1) HUPR_PAIR(6) HUPR_PAIR(7) HUPR_SB();                                                                 \
        HUPR_SM(2, 0) HUPR_PAIR(8) HUPR_PAIR(9) HUPR_SB();                                                                 \
        HUPR_SM(2, 1) HUPR_PAIR(10) HUPR_PAIR(11) HUPR_SB();                                                               \
        HUPR_SM(3, 0) HUPR_PAIR(12) HUPR_PAIR(13) HUPR_SB();                                                               \
        HUPR_SM(3, 1) HUPR_PAIR(14) HUPR_PAIR(15)                                                                          \
        /* (the empty asm statements keep LLVM from sinking the exponentials into the NEXT iteration, next to the MFMAs that */ \
        /* consume them: seen twice)                                                                                      */ \
        _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                                      \
            _Pragma("unroll") for (int u = 0; u < 2; ++u) asm volatile("" : "+v"(pb[CUR__][t][u]));                        \
        asm volatile("" : "+v"(sum2[0]), "+v"(sum2[1]));                                                                   \
        HUPR_SB();                                                                                                         \
        /* ---- tail: the V fragments of tile j for the next iteration, the row sum, the (rare) rescale ---- */            \
        { HUPR_PP_READ_V(j & (kPPRing - 1)) }                                                                     \
        sum2[0] += sum2[1];                                                                                                \
        if constexpr (QS) l_run += sum2[0].x + sum2[0].y;                                                                  \
        else l_run = l_run * alpha + (sum2[0].x + sum2[0].y);                                                              \
        m_run = m_new;                                                                                                     \
        if (!QS && __builtin_amdgcn_ballot_w64(alpha != 1.f)) { /* rare after the first tiles; behind the MFMAs of tile j - 1 */ \
            _Pragma("unroll") for (int ct = 0; ct < 2; ++ct)                                                               \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;                                          \
        }                                                                                                                  \
        HUPR_PP_STAMP()                                                                                                    \
        {                                                                                                                \
            HUPR_PP_VMCNT(4);                                 /* this wave's pieces of tile j + 2 have landed */           \
            HUPR_PP_BARRIER();                                                                                             \
        }                                                                                                                  \
    }
    int j = 0;
    HUPR_PP_TILE(0, 1, true, false)
    for (j = 1; j < nt - 1; j += 2) {
        HUPR_PP_TILE(1, 0, false, false)
        ++j;
        HUPR_PP_TILE(0, 1, false, false)
        --j;
    }
    j = nt - 1;
    HUPR_PP_TILE(1, 0, false, true)
#undef HUPR_PP_TILE
    {                                                         // the last tile's O^T += V^T P^T (its V fragments were read in its tail)
        constexpr int NXT__ = 1;
        constexpr bool FIRST__ = false;
        HUPR_PVM(0, 0, 0) HUPR_PVM(0, 0, 1) HUPR_PVM(0, 1, 0) HUPR_PVM(0, 1, 1)
        HUPR_PVM(1, 0, 0) HUPR_PVM(1, 0, 1) HUPR_PVM(1, 1, 0) HUPR_PVM(1, 1, 1)
    }
#undef HUPR_PVM
#undef HUPR_SM
#undef HUPR_M3
#undef HUPR_PAIR
#undef HUPR_SB
    HUPR_PP_VMCNT(0);
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    {
    store_ct<D>(out + base + (long)q * D, o, 1.f / l_tot, Vres ? Vres + base + (long)q * D : nullptr, lh);
    if (out16)
        store_ct16<D>(out16 + ((long)b * N + q) * ld16, o, 1.f / l_tot, Vres ? Vres + base + (long)q * D : nullptr, lh);
    }
    if (lh == 0) lse[(long)b * N + q] = (QS ? (m_run + __log2f(l_tot)) * kLn2 : m_run + __logf(l_tot));
#undef HUPR_PP_ISSUE
#undef HUPR_PP_READ_K
#undef HUPR_PP_READ_V
#undef HUPR_PP_S
#undef HUPR_PP_PV
#undef HUPR_PP_STAMP
}

}  // namespace hupr

using namespace hupr;

extern "C" int hupr_attn_flash_supported(int N, int C) { return ((C == 64 || C == 128 || C == 256) && N % 128 == 0 && N >= 128) ? 1 : 0; }

// key shares of the split forward: only when the plain grid (N / 128 x Bn workgroups) leaves more than half of the 256 CUs idle;
// then enough shares for ~512 workgroups, a power of two, at least one 64-key tile each
// Policy: single-sample inference only by default (Bn == 1: every level of the decoder is then far below one workgroup per
// CU).  Larger batches keep the one-pass kernel and with it the exact bf16 rounding the parity gates of the training
// configurations were measured with (the shares round P relative to their own running maxima); hupr_debug_attn_split(1)
// widens it to every grid below 128 workgroups, -1 switches it off.
static int g_attn_split = 0;
extern "C" void hupr_debug_attn_split(int mode) { g_attn_split = mode; }
static unsigned long long* g_attn_trace = nullptr;      // profiling: device buffer of 3 x 2 x 4096 s_memtime stamps (fwd, dQ, dK/dV) or null
extern "C" void hupr_debug_attn_trace(void* buf) { g_attn_trace = static_cast<unsigned long long*>(buf); }
static int attn_splits(int Bn, int N) {
    const long wgs = (long)Bn * (N / 128);
    if (wgs >= 128 || g_attn_split < 0 || (g_attn_split == 0 && Bn != 1)) return 1;
    int S = 1;
    while (S * 2 * wgs <= 512 && (N / 64) % (S * 2) == 0) S *= 2;
    return S;
}
extern "C" size_t hupr_attn_fwd_split_ws_bytes(int Bn, int N, int C) {
    const int S = attn_splits(Bn, N);
    return S > 1 ? (size_t)S * Bn * N * (C + 2) * sizeof(float) : 0;
}

template <typename TI, bool QS = false>
static int attn_fwd(const char* who, const TI* K, int ldk, const TI* Q, int ldq, const TI* V, const float* Vres, float* out,
                    float* lse, void* out16, int ld16, int Bn, int N, int C, hupr_stream_t stream, void* ws = nullptr,
                    size_t ws_bytes = 0) {
    HUPR_REQUIRE(K && Q && V && out && lse && Bn > 0, "%s: bad argument", who);
    HUPR_REQUIRE(hupr_attn_flash_supported(N, C), "%s: unsupported shape N=%d C=%d", who, N, C);
    HUPR_REQUIRE(ldk >= C && ldq >= C && ldk % 8 == 0 && ldq % 8 == 0, "%s: bad row strides %d %d", who, ldk, ldq);
    HUPR_REQUIRE(!out16 || (ld16 >= C && ld16 % 4 == 0), "%s: bad bf16 output stride %d", who, ld16);
    dim3 grid(N / 128, Bn);
    __bf16* o16 = static_cast<__bf16*>(out16);
    float* const np = nullptr;
    const int S = ws ? attn_splits(Bn, N) : 1;
    if constexpr (sizeof(TI) == 2) {
        // level-1 shape: the ping-pong kernel (256 queries per 512-thread workgroup, LDS-DMA ring)
        if (S <= 1 && C == 64 && N % 256 == 0 && N >= 256 && (long)N * ldk * 2 < (1L << 31)) {
#define HUPR_PP_FWD() HUPR_LAUNCH(hupr_k_attn_fwd_pp64<false>, dim3((N / 256) * Bn), dim3(512), 0, as_stream(stream), K, Q, V, Vres, \
                                           out, lse, N, Bn, ldk, ldq, o16, ld16, g_attn_trace)
            if constexpr (QS) {
                HUPR_LAUNCH((hupr_k_attn_fwd_pp64<true>), dim3((N / 256) * Bn), dim3(512), 0, as_stream(stream), K, Q, V, Vres, out, lse,
                            N, Bn, ldk, ldq, o16, ld16, g_attn_trace);
                HUPR_LAUNCH_OK("hupr_k_attn_fwd_pp64 (QS)");
                return HUPR_OK;
            }
            HUPR_PP_FWD();
#undef HUPR_PP_FWD
            HUPR_LAUNCH_OK("hupr_k_attn_fwd_pp64");
            return HUPR_OK;
        }
    }
    if (S > 1) {
        HUPR_REQUIRE(ws_bytes >= hupr_attn_fwd_split_ws_bytes(Bn, N, C), "%s: workspace too small", who);
        const long rows = (long)Bn * N;
        float* part_o = static_cast<float*>(ws);
        float* part_ml = part_o + (long)S * rows * C;
        grid.z = S;
        const dim3 cgrid((unsigned)((rows * (C / 4) + 255) / 256));
        hipStream_t s = as_stream(stream);
#define HUPR_ATTN_SPLIT(D_)                                                                                                \
        HUPR_LAUNCH((hupr_k_attn_fwd<D_, TI, true, QS>), grid, dim3(256), 0, s, K, Q, V, Vres, out, lse, N, ldk, ldq, o16, ld16, \
                           part_o, part_ml, AttnBatch());                                                                  \
        HUPR_LAUNCH((hupr_k_attn_combine<D_, QS>), cgrid, dim3(256), 0, s, part_o, part_ml, S, rows, Vres, out, lse, o16, ld16, AttnBatch());
        if (C == 64) { HUPR_ATTN_SPLIT(64) } else if (C == 128) { HUPR_ATTN_SPLIT(128) } else { HUPR_ATTN_SPLIT(256) }
#undef HUPR_ATTN_SPLIT
        HUPR_LAUNCH_OK("hupr_k_attn_fwd (split)");
        return HUPR_OK;
    }
    if (C == 64) HUPR_LAUNCH((hupr_k_attn_fwd<64, TI, false, QS>), grid, dim3(256), 0, as_stream(stream), K, Q, V, Vres, out, lse, N, ldk, ldq, o16, ld16, np, np, AttnBatch());
    else if (C == 128) HUPR_LAUNCH((hupr_k_attn_fwd<128, TI, false, QS>), grid, dim3(256), 0, as_stream(stream), K, Q, V, Vres, out, lse, N, ldk, ldq, o16, ld16, np, np, AttnBatch());
    else HUPR_LAUNCH((hupr_k_attn_fwd<256, TI, false, QS>), grid, dim3(256), 0, as_stream(stream), K, Q, V, Vres, out, lse, N, ldk, ldq, o16, ld16, np, np, AttnBatch());
    HUPR_LAUNCH_OK("hupr_k_attn_fwd");
    return HUPR_OK;
}

// out (B,N,C) = softmax_keys(K Q^T)-weighted V (+V); lse (B,N) saved for the backward
extern "C" int hupr_attn_fwd_bf16(const float* K, const float* Q, const float* V, float* out, float* lse, int Bn, int N, int C,
                                  int residual, hupr_stream_t stream) {
    return attn_fwd("hupr_attn_fwd_bf16", K, C, Q, C, V, residual ? V : nullptr, out, lse, nullptr, 0, Bn, N, C, stream);
}
// same with K, Q, V given as pre-rounded bf16 copies (hupr_cast_f32_to_bf16); Vres: fp32 V for the residual, or null
extern "C" int hupr_attn_fwd_bf16in(const void* K, const void* Q, const void* V, const float* Vres, float* out, float* lse,
                                    int Bn, int N, int C, hupr_stream_t stream) {
    return attn_fwd("hupr_attn_fwd_bf16in", static_cast<const __bf16*>(K), C, static_cast<const __bf16*>(Q), C,
                    static_cast<const __bf16*>(V), Vres, out, lse, nullptr, 0, Bn, N, C, stream);
}
// ... and with row strides ldk / ldq (elements) for K and Q: the key / query projections of one map stored side by side
// in one (B, N, ld) tensor (MSCSA level: four 1x1 projections of a map computed by one GEMM); out16 (optional): a bf16
// copy of the output written with row stride ld16 (a column block of the decoder's concatenated input)
extern "C" int hupr_attn_fwd_bf16in_ld(const void* K, int ldk, const void* Q, int ldq, const void* V, const float* Vres,
                                       float* out, float* lse, void* out16, int ld16, int Bn, int N, int C,
                                       hupr_stream_t stream) {
    return attn_fwd("hupr_attn_fwd_bf16in_ld", static_cast<const __bf16*>(K), ldk, static_cast<const __bf16*>(Q), ldq,
                    static_cast<const __bf16*>(V), Vres, out, lse, out16, ld16, Bn, N, C, stream);
}
// Up to four independent attentions of the same shape and strides (the four of an MSCSA level) in as few launches as fill the chip:
// single-sample inference (hupr_attn_fwd_split_ws_bytes(Bn, N, C) > 0; ws: n_items times that many bytes): ONE split launch + ONE merge
// launch instead of 8; training batches (ws may be null): ONE launch of the one-pass kernel with blockIdx.z = item (levels 2 and 3: a
// single attention's grid is 256 / 64 workgroups), except at the level-1 shape, whose ping-pong kernel is launched once per item.  items: host array of hupr_attn_item (bf16 K / Q / V, fp32 Vres or null, fp32 out, lse, bf16 out16 or null).
template <bool QS>
static int attn_fwd_batch(const char* who, const hupr_attn_item* items, int n_items, int ldk, int ldq, int ld16, int Bn, int N,
                          int C, void* ws, size_t ws_bytes, hupr_stream_t stream) {
    HUPR_REQUIRE(items && n_items >= 1 && n_items <= 4 && Bn > 0, "%s: bad argument", who);
    HUPR_REQUIRE(hupr_attn_flash_supported(N, C), "%s: unsupported shape N=%d C=%d", who, N, C);
    HUPR_REQUIRE(ldk >= C && ldq >= C && ldk % 8 == 0 && ldq % 8 == 0, "%s: bad row strides %d %d", who, ldk, ldq);
    const int S = ws ? attn_splits(Bn, N) : 1;
    if (S == 1 && C == 64 && N % 256 == 0 && (long)N * ldk * 2 < (1L << 31)) {
        // level-1 shape: each attention fills the chip by itself with the ping-pong kernel — one launch per item
        for (int i = 0; i < n_items; ++i) {
            const int rc = attn_fwd<__bf16, QS>(who, static_cast<const __bf16*>(items[i].K), ldk, static_cast<const __bf16*>(items[i].Q), ldq,
                                               static_cast<const __bf16*>(items[i].V), items[i].Vres, items[i].out, items[i].lse,
                                               items[i].out16, ld16, Bn, N, C, stream);
            if (rc) return rc;
        }
        return HUPR_OK;
    }
    HUPR_REQUIRE(S == 1 || ws_bytes >= (size_t)n_items * hupr_attn_fwd_split_ws_bytes(Bn, N, C), "%s: workspace too small", who);
    AttnBatch b = AttnBatch();
    b.n = n_items;
    b.splits = S;
    bool any16 = false;
    for (int i = 0; i < n_items; ++i) {
        HUPR_REQUIRE(items[i].K && items[i].Q && items[i].V && items[i].out && items[i].lse, "%s: null pointer in item %d", who, i);
        b.K[i] = items[i].K; b.Q[i] = items[i].Q; b.V[i] = items[i].V; b.Vres[i] = items[i].Vres;
        b.out[i] = items[i].out; b.lse[i] = items[i].lse; b.out16[i] = static_cast<__bf16*>(items[i].out16);
        any16 = any16 || items[i].out16;
    }
    HUPR_REQUIRE(!any16 || (ld16 >= C && ld16 % 4 == 0), "%s: bad bf16 output stride %d", who, ld16);
    if (S == 1) {                                   // one-pass kernel, blockIdx.z = item
        const dim3 g1(N / 128, Bn, n_items);
        hipStream_t s1 = as_stream(stream);
        const __bf16* const k0 = nullptr;
        float* const f0 = nullptr;
        __bf16* const h0 = nullptr;
        if (C == 64) HUPR_LAUNCH((hupr_k_attn_fwd<64, __bf16, false, QS>), g1, dim3(256), 0, s1, k0, k0, k0, f0, f0, f0, N, ldk, ldq, h0, ld16, f0, f0, b);
        else if (C == 128) HUPR_LAUNCH((hupr_k_attn_fwd<128, __bf16, false, QS>), g1, dim3(256), 0, s1, k0, k0, k0, f0, f0, f0, N, ldk, ldq, h0, ld16, f0, f0, b);
        else HUPR_LAUNCH((hupr_k_attn_fwd<256, __bf16, false, QS>), g1, dim3(256), 0, s1, k0, k0, k0, f0, f0, f0, N, ldk, ldq, h0, ld16, f0, f0, b);
        HUPR_LAUNCH_OK("hupr_k_attn_fwd (batch)");
        return HUPR_OK;
    }
    const long rows = (long)Bn * N;
    float* part_o = static_cast<float*>(ws);
    float* part_ml = part_o + (long)n_items * S * rows * C;
    const dim3 grid(N / 128, Bn, S * n_items);
    const dim3 cgrid((unsigned)((rows * (C / 4) + 255) / 256), n_items);
    hipStream_t s = as_stream(stream);
    typedef __bf16 TI;
    const TI* const nk = nullptr;
    float* const nf = nullptr;
    __bf16* const nh = nullptr;
#define HUPR_ATTN_BATCH(D_)                                                                                                          \
    HUPR_LAUNCH((hupr_k_attn_fwd<D_, TI, true, QS>), grid, dim3(256), 0, s, nk, nk, nk, nf, nf, nf, N, ldk, ldq, nh, ld16, part_o, part_ml, b); \
    HUPR_LAUNCH((hupr_k_attn_combine<D_, QS>), cgrid, dim3(256), 0, s, part_o, part_ml, S, rows, nf, nf, nf, nh, ld16, b);
    if (C == 64) { HUPR_ATTN_BATCH(64) } else if (C == 128) { HUPR_ATTN_BATCH(128) } else { HUPR_ATTN_BATCH(256) }
#undef HUPR_ATTN_BATCH
    HUPR_LAUNCH_OK("hupr_k_attn_fwd (split, batch)");
    return HUPR_OK;
}
extern "C" int hupr_attn_fwd_bf16in_ld_ws_batch(const hupr_attn_item* items, int n_items, int ldk, int ldq, int ld16, int Bn, int N,
                                                int C, void* ws, size_t ws_bytes, hupr_stream_t stream) {
    return attn_fwd_batch<false>("hupr_attn_fwd_bf16in_ld_ws_batch", items, n_items, ldk, ldq, ld16, Bn, N, C, ws, ws_bytes, stream);
}
// ... with Q' = log2(e) Q handed over (the QS kernels; see kDeferBits above)
extern "C" int hupr_attn_fwd_bf16in_ld_ws_batch_qs(const hupr_attn_item* items, int n_items, int ldk, int ldq, int ld16, int Bn, int N,
                                                   int C, void* ws, size_t ws_bytes, hupr_stream_t stream) {
    return attn_fwd_batch<true>("hupr_attn_fwd_bf16in_ld_ws_batch_qs", items, n_items, ldk, ldq, ld16, Bn, N, C, ws, ws_bytes, stream);
}

// the same with a workspace of hupr_attn_fwd_split_ws_bytes(Bn, N, C) bytes (0: the plain kernel already fills the GPU and ws may
// be null): small batches split the keys over blockIdx.z and merge the shares in a second launch (flash-decoding)
extern "C" int hupr_attn_fwd_bf16in_ld_ws(const void* K, int ldk, const void* Q, int ldq, const void* V, const float* Vres,
                                          float* out, float* lse, void* out16, int ld16, int Bn, int N, int C, void* ws,
                                          size_t ws_bytes, hupr_stream_t stream) {
    return attn_fwd("hupr_attn_fwd_bf16in_ld_ws", static_cast<const __bf16*>(K), ldk, static_cast<const __bf16*>(Q), ldq,
                    static_cast<const __bf16*>(V), Vres, out, lse, out16, ld16, Bn, N, C, stream, ws, ws_bytes);
}
// The same attention with the query operand pre-scaled: Q' = log2(e) Q (bf16).  out / lse are those of softmax(K Q^T) — lse in
// natural units, as always.  The level-1 shape runs the ping-pong kernel with the deferred maximum (kDeferBits).
extern "C" int hupr_attn_fwd_bf16in_ld_ws_qs(const void* K, int ldk, const void* Qs, int ldq, const void* V, const float* Vres,
                                             float* out, float* lse, void* out16, int ld16, int Bn, int N, int C, void* ws,
                                             size_t ws_bytes, hupr_stream_t stream) {
    return attn_fwd<__bf16, true>("hupr_attn_fwd_bf16in_ld_ws_qs", static_cast<const __bf16*>(K), ldk, static_cast<const __bf16*>(Qs), ldq,
                                  static_cast<const __bf16*>(V), Vres, out, lse, out16, ld16, Bn, N, C, stream, ws, ws_bytes);
}

// dK, dQ, dV (B,N,C) from dout; Dq: scratch (B,N) floats.  V32 / out / dout32: fp32 tensors of the exact row-sum
// D = rowsum(dO o (out - V)) and the residual epilogue; K, Q, V, dO: the MFMA operands (fp32 or bf16 copies).
// ldk / ldq / lddk / lddq / lddo: row strides of K, Q, dK, dQ, dO.  dVadd: tensor added to dV in the epilogue (dout32 for
// the residual form; may be dV itself to accumulate onto what another attention over the same values left there), or
// null.  dout32 == null (bf16 dO only): the gradient arrived bf16-stored, so dO itself is the exact gradient — the
// row-sum and the residual term read it directly.
template <typename TI, bool QS = false>
static int attn_bwd(const char* who, const TI* K, int ldk, const TI* Q, int ldq, const TI* V, const TI* dO, int lddo,
                    const float* V32, const float* out, const float* dout32, const float* lse, float* dK, int lddk, float* dQ,
                    int lddq, float* dV, int accumulate, float* Dq, int Bn, int N, int C, int residual, hupr_stream_t stream) {
    HUPR_REQUIRE(K && Q && V && dO && V32 && out && lse && dK && dQ && dV && Dq && Bn > 0, "%s: bad argument", who);
    HUPR_REQUIRE(dout32 || sizeof(TI) == 2, "%s: fp32 operands need the fp32 gradient", who);
    HUPR_REQUIRE(hupr_attn_flash_supported(N, C), "%s: unsupported shape N=%d C=%d", who, N, C);
    HUPR_REQUIRE(ldk >= C && ldq >= C && lddk >= C && lddq >= C && lddo >= C && ldk % 8 == 0 && ldq % 8 == 0 && lddo % 8 == 0 &&
                     lddk % 4 == 0 && lddq % 4 == 0,
                 "%s: bad row strides", who);
    HUPR_REQUIRE(!(residual && accumulate), "%s: accumulate is for the non-residual form", who);
    hipStream_t s = as_stream(stream);
    const long rows = (long)Bn * N;
    dim3 grid(N / 128, Bn);
    const float* add32 = residual ? dout32 : (accumulate ? dV : nullptr);
    const __bf16* add16 = (residual && !dout32) ? reinterpret_cast<const __bf16*>(dO) : nullptr;
    const dim3 pgrid((unsigned)((rows + 15) / 16));
    const int xmap = (Bn % 8 == 0) ? 1 : 0;
#define HUPR_ATTN_BWD(D_, NH_)                                                                                             \
    if (dout32) HUPR_LAUNCH((hupr_k_attn_prep<D_, float>), pgrid, dim3(256), 0, s, dout32, C, out, V32, Dq, rows, residual); \
    else HUPR_LAUNCH((hupr_k_attn_prep<D_, __bf16>), pgrid, dim3(256), 0, s, reinterpret_cast<const __bf16*>(dO), lddo, out, V32, Dq, rows, residual); \
    HUPR_LAUNCH((hupr_k_attn_bwd_dq<D_, TI, QS>), grid, dim3(256), 0, s, K, Q, V, dO, lse, Dq, dQ, N, ldk, ldq, lddq, lddo, xmap);  \
    if (D_ == 64 && sizeof(TI) == 2 && N % 256 == 0)                                                       \
        HUPR_LAUNCH(hupr_k_attn_bwd_dkv512<QS>, dim3(N / 256, Bn), dim3(512), 0, s, reinterpret_cast<const __bf16*>(K),       \
                           reinterpret_cast<const __bf16*>(Q), reinterpret_cast<const __bf16*>(V), reinterpret_cast<const __bf16*>(dO), \
                           add32, lse, Dq, dK, dV, N, ldk, ldq, lddk, lddo, add16, lddo, xmap);                                 \
    else                                                                                                                     \
        HUPR_LAUNCH((hupr_k_attn_bwd_dkv<D_, TI, NH_, QS>), dim3(N / 128, Bn, NH_), dim3(256), 0, s, K, Q, V, dO, add32, lse, Dq, \
                           dK, dV, N, ldk, ldq, lddk, lddo, add16, lddo, xmap);
    if (C == 64) { HUPR_ATTN_BWD(64, 1) } else if (C == 128) { HUPR_ATTN_BWD(128, 1) } else { HUPR_ATTN_BWD(256, 2) }
#undef HUPR_ATTN_BWD
    HUPR_LAUNCH_OK("hupr_k_attn_bwd");
    return HUPR_OK;
}

// The backward passes of up to four attentions of one shape (an MSCSA level; bf16 operands and a bf16-stored gradient dO): ONE row-sum
// launch, ONE dQ launch, and dK / dV launches in as many rounds as the items' dV targets need — an item with `accumulate` adds onto
// the dV an EARLIER item of the array writes, so it goes into a later round (SPEC order of the level: two rounds of two).  At the
// level-1 shape the 512-thread dK / dV kernel takes one attention per launch (it fills the chip alone), in array order.
template <bool QS>
static int attn_bwd_batch(const char* who, const hupr_attn_bwd_item* items, int n_items, int ldk, int ldq, int lddo, int lddk, int lddq,
                          int Bn, int N, int C, hupr_stream_t stream) {
    HUPR_REQUIRE(items && n_items >= 1 && n_items <= 4 && Bn > 0, "%s: bad argument", who);
    HUPR_REQUIRE(hupr_attn_flash_supported(N, C), "%s: unsupported shape N=%d C=%d", who, N, C);
    for (int i = 0; i < n_items; ++i) {
        const hupr_attn_bwd_item& t = items[i];
        HUPR_REQUIRE(t.K && t.Q && t.V && t.dO && t.V32 && t.out && t.lse && t.dK && t.dQ && t.dV && t.Dq, "%s: null pointer in item %d", who, i);
        HUPR_REQUIRE(!(t.residual && t.accumulate), "%s: accumulate is for the non-residual form (item %d)", who, i);
    }
    const bool level1 = C == 64 && N % 256 == 0;       // its 512-thread dK / dV kernel takes one attention per launch
    if (n_items == 1) {
        for (int i = 0; i < n_items; ++i) {
            const hupr_attn_bwd_item& t = items[i];
            const int rc = attn_bwd<__bf16, QS>(who, static_cast<const __bf16*>(t.K), ldk, static_cast<const __bf16*>(t.Q), ldq,
                                               static_cast<const __bf16*>(t.V), static_cast<const __bf16*>(t.dO), lddo, t.V32, t.out, nullptr,
                                               t.lse, t.dK, lddk, t.dQ, lddq, t.dV, t.accumulate, t.Dq, Bn, N, C, t.residual, stream);
            if (rc) return rc;
        }
        return HUPR_OK;
    }
    HUPR_REQUIRE(ldk >= C && ldq >= C && lddk >= C && lddq >= C && lddo >= C && ldk % 8 == 0 && ldq % 8 == 0 && lddo % 8 == 0 &&
                     lddk % 4 == 0 && lddq % 4 == 0, "%s: bad row strides", who);
    hipStream_t s = as_stream(stream);
    const long rows = (long)Bn * N;
    const int xmap = (Bn % 8 == 0) ? 1 : 0;
    AttnBwdBatch b = AttnBwdBatch();
    b.n = n_items;
    b.Bn = Bn;
    for (int i = 0; i < n_items; ++i) {
        const hupr_attn_bwd_item& t = items[i];
        b.K[i] = t.K; b.Q[i] = t.Q; b.V[i] = t.V; b.dO[i] = t.dO; b.out[i] = t.out; b.V32[i] = t.V32; b.lse[i] = t.lse;
        b.Dq[i] = t.Dq; b.dK[i] = t.dK; b.dQ[i] = t.dQ; b.dV[i] = t.dV; b.residual[i] = t.residual;
        b.add32[i] = t.accumulate ? t.dV : nullptr;
        b.add16[i] = t.residual ? static_cast<const __bf16*>(t.dO) : nullptr;
    }
    const __bf16* const k0 = nullptr;
    const float* const c0 = nullptr;
    float* const f0 = nullptr;
    const dim3 pgrid((unsigned)((rows + 15) / 16), n_items), qgrid(N / 128, Bn * n_items);
    // dK / dV rounds: an item waits for the round after the one that writes the dV it accumulates onto / shares
    int round_of[4], n_rounds = 0;
    for (int i = 0; i < n_items; ++i) {
        int r = 0;
        for (int j = 0; j < i; ++j)
            if (items[j].dV == items[i].dV) r = round_of[j] + 1 > r ? round_of[j] + 1 : r;
        round_of[i] = r;
        n_rounds = r + 1 > n_rounds ? r + 1 : n_rounds;
    }
#define HUPR_ATTN_BWD_BATCH(D_, NH_)                                                                                             \
    HUPR_LAUNCH((hupr_k_attn_prep<D_, __bf16>), pgrid, dim3(256), 0, s, k0, lddo, c0, c0, f0, rows, 0, b);                      \
    HUPR_LAUNCH((hupr_k_attn_bwd_dq<D_, __bf16, QS>), qgrid, dim3(256), 0, s, k0, k0, k0, k0, c0, c0, f0, N, ldk, ldq, lddq, lddo, xmap, b); \
    for (int r = 0; r < n_rounds; ++r) {                                                                                         \
        AttnBwdBatch br = AttnBwdBatch();                                                                                        \
        br.Bn = Bn;                                                                                                              \
        for (int i = 0; i < n_items; ++i)                                                                                        \
            if (round_of[i] == r) {                                                                                              \
                const int k = br.n++;                                                                                            \
                br.K[k] = b.K[i]; br.Q[k] = b.Q[i]; br.V[k] = b.V[i]; br.dO[k] = b.dO[i]; br.lse[k] = b.lse[i]; br.Dq[k] = b.Dq[i]; \
                br.dK[k] = b.dK[i]; br.dV[k] = b.dV[i]; br.add32[k] = b.add32[i]; br.add16[k] = b.add16[i];                      \
            }                                                                                                                    \
        HUPR_LAUNCH((hupr_k_attn_bwd_dkv<D_, __bf16, NH_, QS>), dim3(N / 128, Bn * br.n, NH_), dim3(256), 0, s, k0, k0, k0, k0, c0, c0, c0, \
                    f0, f0, N, ldk, ldq, lddk, lddo, k0, lddo, xmap, br);                                                        \
    }
    if (level1) {
        HUPR_LAUNCH((hupr_k_attn_prep<64, __bf16>), pgrid, dim3(256), 0, s, k0, lddo, c0, c0, f0, rows, 0, b);
        HUPR_LAUNCH((hupr_k_attn_bwd_dq<64, __bf16, QS>), qgrid, dim3(256), 0, s, k0, k0, k0, k0, c0, c0, f0, N, ldk, ldq, lddq, lddo, xmap, b);
        for (int i = 0; i < n_items; ++i)       // array order: an accumulating item follows the one that writes its dV
            HUPR_LAUNCH(hupr_k_attn_bwd_dkv512<QS>, dim3(N / 256, Bn), dim3(512), 0, s, static_cast<const __bf16*>(b.K[i]),
                        static_cast<const __bf16*>(b.Q[i]), static_cast<const __bf16*>(b.V[i]), static_cast<const __bf16*>(b.dO[i]), b.add32[i],
                        b.lse[i], b.Dq[i], b.dK[i], b.dV[i], N, ldk, ldq, lddk, lddo, b.add16[i], lddo, xmap);
    } else if (C == 64) { HUPR_ATTN_BWD_BATCH(64, 1) } else if (C == 128) { HUPR_ATTN_BWD_BATCH(128, 1) } else { HUPR_ATTN_BWD_BATCH(256, 2) }
#undef HUPR_ATTN_BWD_BATCH
    HUPR_LAUNCH_OK("hupr_k_attn_bwd (batch)");
    return HUPR_OK;
}
extern "C" int hupr_attn_bwd_bf16in_ld_batch(const hupr_attn_bwd_item* items, int n_items, int ldk, int ldq, int lddo, int lddk,
                                             int lddq, int Bn, int N, int C, hupr_stream_t stream) {
    return attn_bwd_batch<false>("hupr_attn_bwd_bf16in_ld_batch", items, n_items, ldk, ldq, lddo, lddk, lddq, Bn, N, C, stream);
}
extern "C" int hupr_attn_bwd_bf16in_ld_batch_qs(const hupr_attn_bwd_item* items, int n_items, int ldk, int ldq, int lddo, int lddk,
                                                int lddq, int Bn, int N, int C, hupr_stream_t stream) {
    return attn_bwd_batch<true>("hupr_attn_bwd_bf16in_ld_batch_qs", items, n_items, ldk, ldq, lddo, lddk, lddq, Bn, N, C, stream);
}

extern "C" int hupr_attn_bwd_bf16(const float* K, const float* Q, const float* V, const float* out, const float* dout,
                                  const float* lse, float* dK, float* dQ, float* dV, float* Dq, int Bn, int N, int C,
                                  int residual, hupr_stream_t stream) {
    return attn_bwd("hupr_attn_bwd_bf16", K, C, Q, C, V, dout, C, V, out, dout, lse, dK, C, dQ, C, dV, 0, Dq, Bn, N, C, residual,
                    stream);
}
extern "C" int hupr_attn_bwd_bf16in(const void* K, const void* Q, const void* V, const void* dO, const float* V32,
                                    const float* out, const float* dout32, const float* lse, float* dK, float* dQ, float* dV,
                                    float* Dq, int Bn, int N, int C, int residual, hupr_stream_t stream) {
    HUPR_REQUIRE(dout32, "hupr_attn_bwd_bf16in: null pointer");
    return attn_bwd("hupr_attn_bwd_bf16in", static_cast<const __bf16*>(K), C, static_cast<const __bf16*>(Q), C,
                    static_cast<const __bf16*>(V), static_cast<const __bf16*>(dO), C, V32, out, dout32, lse, dK, C, dQ, C, dV, 0,
                    Dq, Bn, N, C, residual, stream);
}
// strided form (see hupr_attn_fwd_bf16in_ld): dK / dQ land in column blocks of wider gradient tensors; dO has row
// stride lddo (a column block of the gradient of the decoder's concatenated input); dout32 may be null (see above);
// accumulate != 0 (non-residual form only) adds the result onto the dV already in place
extern "C" int hupr_attn_bwd_bf16in_ld(const void* K, int ldk, const void* Q, int ldq, const void* V, const void* dO, int lddo,
                                       const float* V32, const float* out, const float* dout32, const float* lse, float* dK,
                                       int lddk, float* dQ, int lddq, float* dV, float* Dq, int Bn, int N, int C,
                                       int residual, int accumulate, hupr_stream_t stream) {
    return attn_bwd("hupr_attn_bwd_bf16in_ld", static_cast<const __bf16*>(K), ldk, static_cast<const __bf16*>(Q), ldq,
                    static_cast<const __bf16*>(V), static_cast<const __bf16*>(dO), lddo, V32, out, dout32, lse, dK, lddk, dQ,
                    lddq, dV, accumulate, Dq, Bn, N, C, residual, stream);
}
// ... with Q' = log2(e) Q as the query operand (what hupr_attn_fwd_bf16in_ld_ws_qs was given): dK, dQ, dV are the gradients with
// respect to K, the UNSCALED Q and V, as above (the kernels sum dK over Q' and rescale it by ln 2 on the way out)
extern "C" int hupr_attn_bwd_bf16in_ld_qs(const void* K, int ldk, const void* Qs, int ldq, const void* V, const void* dO, int lddo,
                                          const float* V32, const float* out, const float* dout32, const float* lse, float* dK,
                                          int lddk, float* dQ, int lddq, float* dV, float* Dq, int Bn, int N, int C,
                                          int residual, int accumulate, hupr_stream_t stream) {
    return attn_bwd<__bf16, true>("hupr_attn_bwd_bf16in_ld_qs", static_cast<const __bf16*>(K), ldk, static_cast<const __bf16*>(Qs), ldq,
                                  static_cast<const __bf16*>(V), static_cast<const __bf16*>(dO), lddo, V32, out, dout32, lse, dK, lddk,
                                  dQ, lddq, dV, accumulate, Dq, Bn, N, C, residual, stream);
}
