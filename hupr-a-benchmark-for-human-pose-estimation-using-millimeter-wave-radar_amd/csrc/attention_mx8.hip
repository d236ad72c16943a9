// BASELINE.json config 5, second form: MSCSA attention forward on the block-scaled fp8 matrix instruction of gfx950
// (v_mfma_scale_f32_32x32x64_f8f6f4: OCP e4m3 operands, one E8M0 power-of-two scale per 32 elements of the reduction axis,
// applied inside the matrix pipe; twice the bf16 rate, half the operand bytes) — for the level that carries 88 % of the
// attention flops (C = 64 channels, N = 4096 tokens).  The round-2 kernel (attention_fp8.hip) used ONE scale per tensor and
// lost decoded-head agreement; the scheme study of round 4 (profiles/r04_attn_fp8_schemes.txt) put 32-element blocks with
// power-of-two scales and probabilities stored as 2^8 p at the bf16 path's own level.  This file builds that scheme.
//
// Reference semantics (models/layers.py:126-133): S[j,q] = sum_c K[j,c] Q[q,c]; P = softmax over keys j;
// out[q,c] = sum_j P[j,q] V[j,c] (+ V[q,c]).  Same keys-x-queries orientation and online softmax as attention_bf16.hip.
//
// Operand layout of the instruction, determined by experiment (scripts/probes/mfma_scale_probe.hip, profiles/r04_mfma_scale_probe.txt;
// 32 x 32 x 64, lane l: row / column l & 31, half h = l >> 5, 32 bytes = 8 VGPRs): byte m of half h is reduction element
// 32 (m >> 4) + 16 h + (m & 15) — the two 32-element SCALE blocks are bytes 0..15 of both halves and bytes 16..31 of both
// halves — and the scale byte in lane (row, h) multiplies block h (so a lane's scale covers half of its own bytes and half of its
// partner's; a first build that gave each lane the scale of "its" 32 bytes was wrong by powers of two wherever the two blocks
// of a row differ).  Hence:
//   S^T = K Q^T:   A = 32 keys x 64 channels: lane (key, h) reads the 16-byte chunks h and 2 + h of its key's 64-byte row (block 0 =
//                  channels 0..31, block 1 = 32..63) and supplies the scale of channel block h; B = the lane's query row, same
//                  chunks, in registers; ONE instruction per 32 x 32 score tile, the scales cost nothing;
//   O^T += V^T P^T: the reduction runs over 64 keys.  A probability tile pair in its accumulator layout is ALREADY a B operand:
//                  byte m = 16 t + r of half h is key 32 t + 8 (r >> 2) + 4 h + (r & 3), i.e. scale block t = keys 32 t .. 32 t + 31
//                  of the tile; the quantiser writes V TRANSPOSED with the keys of every 64-key tile in exactly that byte order
//                  ([channel][h * 32 + m]) and one scale per (channel, 32 consecutive keys), so the A operand is a plain 32-byte
//                  row read and lane (channel, h) supplies the scale of keys 32 h .. 32 h + 31.
// Quantisation (hupr_attn_mx8_quant_level): one pass over the level's two projection tensors (B, N, 4 C) and its two value
// maps, bf16 in, bytes + scale bytes out.  Block scale 2^e with e = ceil(log2(amax / 448)): nothing saturates.
#include "gemm_common.h"

namespace hupr {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4m __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8m __attribute__((ext_vector_type(8)));

constexpr float kLog2eM = 1.4426950408889634f;

__device__ __forceinline__ unsigned pack4_e4m3_m(float a, float b, float c, float d) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}

// E8M0 byte of the block scale 2^e, e = ceil(log2(amax / 448)) (127 = 1.0 for an all-zero block), and the exact multiplier 2^-e
__device__ __forceinline__ void mx_scale(float amax, unsigned& byte, float& mul) {
    const unsigned bits = __float_as_uint(amax * (1.f / 448.f));
    unsigned e = (bits >> 23) & 0xffu;
    if (bits & 0x7fffffu) ++e;
    e = amax > 0.f ? min(max(e, 1u), 253u) : 127u;
    byte = e;
    mul = __uint_as_float((254u - e) << 23);
}

// rows x ncols bf16 (dense) -> e4m3 bytes [rows][ncols] + scale bytes [rows][ncols / 32]; one thread per 32-element block
__global__ __launch_bounds__(256) void hupr_k_quant_rows_mx8(const __bf16* __restrict__ x, unsigned char* __restrict__ y,
                                                             unsigned char* __restrict__ sc, long nblocks) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nblocks; i += (long)gridDim.x * 256) {
        const bf16x8m* src = reinterpret_cast<const bf16x8m*>(x + i * 32);
        bf16x8m v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = src[u];
        float amax = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf((float)v[u][j]));
        unsigned byte;
        float mul;
        mx_scale(amax, byte, mul);
        u32x4m o[2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            o[u >> 1][2 * (u & 1)] = pack4_e4m3_m((float)v[u][0] * mul, (float)v[u][1] * mul, (float)v[u][2] * mul, (float)v[u][3] * mul);
            o[u >> 1][2 * (u & 1) + 1] = pack4_e4m3_m((float)v[u][4] * mul, (float)v[u][5] * mul, (float)v[u][6] * mul, (float)v[u][7] * mul);
        }
        u32x4m* dst = reinterpret_cast<u32x4m*>(y + i * 32);
        dst[0] = o[0];
        dst[1] = o[1];
        sc[i] = (unsigned char)byte;
    }
}

// v (B, N, 64) bf16 -> per sample and 64-key tile: VT8 [64 channels][64 bytes = (h, m)] and scale pairs [64 channels][key block t]
__global__ __launch_bounds__(256) void hupr_k_quant_vt_mx8(const __bf16* __restrict__ v, unsigned char* __restrict__ vt,
                                                           unsigned char* __restrict__ sc, int N) {
    __shared__ __bf16 t[64][66];                              // [key][channel], odd dword pitch
    const int tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const __bf16* src = v + ((long)b * N + (long)tile * 64) * 64;
    for (int i = tid; i < 64 * 8; i += 256) {
        const int r = i >> 3, c8 = i & 7;
        const bf16x8m q = *reinterpret_cast<const bf16x8m*>(src + r * 64 + c8 * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) t[r][c8 * 8 + j] = q[j];
    }
    __syncthreads();
    if (tid < 128) {
        const int c = tid & 63, h = tid >> 6;
        // scale block t = keys 32 t .. 32 t + 31 of the tile (bytes 16 t .. 16 t + 15 of both halves)
        float amax[2] = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 64; ++j) amax[j >> 5] = fmaxf(amax[j >> 5], fabsf((float)t[j][c]));
        unsigned byte[2];
        float mul[2];
        mx_scale(amax[0], byte[0], mul[0]);
        mx_scale(amax[1], byte[1], mul[1]);
        float val[32];
#pragma unroll
        for (int m = 0; m < 32; ++m) val[m] = (float)t[32 * (m >> 4) + 8 * ((m & 15) >> 2) + 4 * h + (m & 3)][c] * mul[m >> 4];
        const long tb = (long)b * (N / 64) + tile;
        u32x4m* dst = reinterpret_cast<u32x4m*>(vt + (tb * 64 + c) * 64 + h * 32);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            u32x4m o;
#pragma unroll
            for (int d = 0; d < 4; ++d)
                o[d] = pack4_e4m3_m(val[16 * u + 4 * d], val[16 * u + 4 * d + 1], val[16 * u + 4 * d + 2], val[16 * u + 4 * d + 3]);
            dst[u] = o;
        }
        sc[(tb * 64 + c) * 2 + h] = (unsigned char)byte[h];
    }
}

// LDS images: 64 rows x 64 bytes; 16-byte chunk c of row r lives at chunk c ^ ((r >> 2) & 3) (a 16-lane ds_read_b128 group covers
// rows of all four (r >> 2) & 3 classes: every bank once)
__device__ __forceinline__ int img16(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// K8 / Q8: e4m3 rows with ldk / ldq BYTES between tokens, the attention's 64 channels at the given pointers; Ksc / Qsc: the scale
// pair of those 64 channels, lsk / lsq bytes between tokens.  VT8 / Vsc as written by hupr_k_quant_vt_mx8.
__global__ __launch_bounds__(256, 2) void hupr_k_attn_fwd_mx8(const unsigned char* __restrict__ K8, int ldk,
                                                              const unsigned char* __restrict__ Ksc, int lsk,
                                                              const unsigned char* __restrict__ Q8, int ldq,
                                                              const unsigned char* __restrict__ Qsc, int lsq,
                                                              const unsigned char* __restrict__ VT8, const unsigned char* __restrict__ Vsc,
                                                              const float* __restrict__ Vres, float* __restrict__ out,
                                                              float* __restrict__ lse, __bf16* __restrict__ out16, int ld16, int N) {
    constexpr int D = 64;
    __shared__ __attribute__((aligned(16))) unsigned char Ks[2][64 * 64];
    __shared__ __attribute__((aligned(16))) unsigned char Vt[2][64 * 64];
    // scale pairs [buffer][row] (byte h = block h): lane (row, h) hands the instruction the scale of block h (one ds_read_u16 + v_bfe;
    // separate byte arrays read with ds_read_u8 measured 13 us slower per call)
    __shared__ unsigned short Kscs[2][64], Vscs[2][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
    const long bN = (long)blockIdx.y * N;
    const int q = blockIdx.x * 128 + wave * 32 + lr;
    // this lane's query: half lh of either channel block, and the scale of block lh
    i32x8 qf;
    {
        const u32x4m* qp = reinterpret_cast<const u32x4m*>(Q8 + (bN + q) * ldq);
        const u32x4m a = qp[lh], b = qp[2 + lh];                    // 16 channels of block 0, 16 of block 1
        qf = (i32x8){(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
    }
    const int qs = Qsc[(bN + q) * lsq + lh];
    f32x16 o[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;                          // l_run in units of 2^8 (the stored probabilities are 256 p)
    // staging: thread t moves 16 bytes of row t >> 2 of each image; threads 0..63 / 64..127 one scale pair each
    const int srow = tid >> 2, sc4 = tid & 3;
    const unsigned char* kp = K8 + (bN + srow) * ldk + sc4 * 16;
    const unsigned char* vp = VT8 + ((long)blockIdx.y * (N / 64) * 64 + srow) * 64 + sc4 * 16;
    const unsigned char* ksp = Ksc + (bN + (tid & 63)) * lsk;
    const unsigned char* vsp = Vsc + ((long)blockIdx.y * (N / 64) * 64 + (tid & 63)) * 2;
    u32x4m kreg = *reinterpret_cast<const u32x4m*>(kp), vreg = *reinterpret_cast<const u32x4m*>(vp);
    unsigned short sreg = tid < 64 ? *reinterpret_cast<const unsigned short*>(ksp)
                                   : (tid < 128 ? *reinterpret_cast<const unsigned short*>(vsp) : (unsigned short)0);
    *reinterpret_cast<u32x4m*>(&Ks[0][img16(srow, sc4)]) = kreg;
    *reinterpret_cast<u32x4m*>(&Vt[0][img16(srow, sc4)]) = vreg;
    if (tid < 64) Kscs[0][tid] = sreg;
    else if (tid < 128) Vscs[0][tid - 64] = sreg;
    __syncthreads();
    const int ntiles = N / 64;
    for (int jt = 0; jt < ntiles; ++jt) {
        const int cb = jt & 1;
        if (jt + 1 < ntiles) {                                     // next tile travels while this one is multiplied
            kreg = *reinterpret_cast<const u32x4m*>(kp + (long)(jt + 1) * 64 * ldk);
            vreg = *reinterpret_cast<const u32x4m*>(vp + (long)(jt + 1) * 64 * 64);
            if (tid < 64) sreg = *reinterpret_cast<const unsigned short*>(ksp + (long)(jt + 1) * 64 * lsk);
            else if (tid < 128) sreg = *reinterpret_cast<const unsigned short*>(vsp + (long)(jt + 1) * 64 * 2);
        }
        // S^T tiles: rows = keys, this lane's column = its query; scales applied by the instruction
        f32x16 st[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const u32x4m a0 = *reinterpret_cast<const u32x4m*>(&Ks[cb][img16(32 * t + lr, lh)]);
            const u32x4m a1 = *reinterpret_cast<const u32x4m*>(&Ks[cb][img16(32 * t + lr, 2 + lh)]);
            const i32x8 ka = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
            const int ks = (Kscs[cb][32 * t + lr] >> (8 * lh)) & 0xff;
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            st[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ka, qf, z, 0, 0, 0, ks, 0, qs);
            // (hipcc also allocates the 16 result registers ON TOP of the A operand when C is the inline constant 0; keeping A
            // and its scale alive past the instruction forces disjoint registers at no cost)
            asm volatile("" ::"v"(ka), "v"(ks));
        }
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[t][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * kLog2eM);
        const float nm = fmaf(-m_new, kLog2eM, 8.f);               // + 8: the probabilities leave the exponential as 256 p
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(st[t][r], kLog2eM, nm));
                st[t][r] = pv;
                sum += pv;
            }
        l_run = l_run * alpha + sum;
        m_run = m_new;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f)) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;
        }
        // the probability tile pair as the B operand: byte m = 16 t + r
        i32x8 pb;
#pragma unroll
        for (int d = 0; d < 8; ++d)
            pb[d] = (int)pack4_e4m3_m(st[d >> 2][4 * (d & 3)], st[d >> 2][4 * (d & 3) + 1], st[d >> 2][4 * (d & 3) + 2],
                                      st[d >> 2][4 * (d & 3) + 3]);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const u32x4m a0 = *reinterpret_cast<const u32x4m*>(&Vt[cb][img16(32 * ct + lr, 2 * lh)]);
            const u32x4m a1 = *reinterpret_cast<const u32x4m*>(&Vt[cb][img16(32 * ct + lr, 2 * lh + 1)]);
            const i32x8 va = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
            const int vs = (Vscs[cb][32 * ct + lr] >> (8 * lh)) & 0xff;
            o[ct] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, pb, o[ct], 0, 0, 0, vs, 0, 127);
        }
        if (jt + 1 < ntiles) {                                     // the other buffer was last read one tile ago, behind a barrier
            *reinterpret_cast<u32x4m*>(&Ks[cb ^ 1][img16(srow, sc4)]) = kreg;
            *reinterpret_cast<u32x4m*>(&Vt[cb ^ 1][img16(srow, sc4)]) = vreg;
            if (tid < 64) Kscs[cb ^ 1][tid] = sreg;
            else if (tid < 128) Vscs[cb ^ 1][tid - 64] = sreg;
        }
        __syncthreads();
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float oscale = 1.f / l_tot;                              // O and l both carry the factor 2^8
    const long base = bN * D;
    float* dst = out + base + (long)q * D;
    const float* add = Vres ? Vres + base + (long)q * D : nullptr;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int c = 32 * ct + 8 * q4 + 4 * lh;
            float4 v = make_float4(o[ct][4 * q4] * oscale, o[ct][4 * q4 + 1] * oscale, o[ct][4 * q4 + 2] * oscale, o[ct][4 * q4 + 3] * oscale);
            if (add) {
                const float4 a = *reinterpret_cast<const float4*>(add + c);
                v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
            }
            *reinterpret_cast<float4*>(dst + c) = v;
            if (out16) {
                typedef __bf16 bf16x4m __attribute__((ext_vector_type(4)));
                const bf16x4m h = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
                *reinterpret_cast<bf16x4m*>(out16 + (bN + q) * ld16 + c) = h;
            }
        }
    if (lh == 0) lse[bN + q] = m_run + __logf(l_tot) - 8.f * 0.6931471805599453f;
}

}  // namespace hupr

using namespace hupr;

// workspace of one MSCSA level: per map the projection bytes (Bn N 4C) + their scale bytes (Bn N 4C / 32), the transposed value
// bytes (Bn N C) + their scale pairs (Bn N / 64 x C x 2); 256-byte aligned pieces
static size_t mx8_piece(size_t n) { return align_up(n, 256); }
struct Mx8Layout {
    size_t y8, ysc, vt8, vsc, per_map;
};
static Mx8Layout mx8_layout(int Bn, int N, int C) {
    Mx8Layout l;
    const size_t rows = (size_t)Bn * N;
    l.y8 = 0;
    l.ysc = l.y8 + mx8_piece(rows * 4 * C);
    l.vt8 = l.ysc + mx8_piece(rows * 4 * C / 32);
    l.vsc = l.vt8 + mx8_piece(rows * C);
    l.per_map = l.vsc + mx8_piece(rows / 64 * C * 2);
    return l;
}

extern "C" size_t hupr_attn_mx8_ws_bytes(int Bn, int N, int C) { return 2 * mx8_layout(Bn, N, C).per_map; }

static int mx8_check(const char* who, int Bn, int N, int C, const void* ws, size_t ws_bytes) {
    HUPR_REQUIRE(ws && Bn > 0, "%s: bad argument", who);
    HUPR_REQUIRE(C == 64 && N % 128 == 0, "%s: only C = 64, N %% 128 == 0 (got C=%d N=%d)", who, C, N);
    if (ws_bytes < hupr_attn_mx8_ws_bytes(Bn, N, C)) return fail(HUPR_ERR_WORKSPACE, "%s: workspace too small", who);
    return HUPR_OK;
}

// step 1: the level's operands -> block-scaled e4m3.  Ya / Ye: bf16 (Bn, N, 4 C) dense (the four projections of a map side by side),
// va / ve: bf16 (Bn, N, C) value maps (map 0 / map 1)
extern "C" int hupr_attn_mx8_quant_level(const void* Ya, const void* Ye, const void* va, const void* ve, int Bn, int N, int C,
                                         void* ws, size_t ws_bytes, hupr_stream_t stream) {
    if (int rc = mx8_check("hupr_attn_mx8_quant_level", Bn, N, C, ws, ws_bytes)) return rc;
    HUPR_REQUIRE(Ya && Ye && va && ve, "hupr_attn_mx8_quant_level: null tensor");
    hipStream_t s = as_stream(stream);
    const Mx8Layout l = mx8_layout(Bn, N, C);
    const long nblocks = (long)Bn * N * 4 * C / 32;
    const void* Y[2] = {Ya, Ye};
    const void* V[2] = {va, ve};
    for (int m = 0; m < 2; ++m) {
        unsigned char* base = static_cast<unsigned char*>(ws) + m * l.per_map;
        HUPR_LAUNCH(hupr_k_quant_rows_mx8, dim3((unsigned)min((long)4096, (nblocks + 255) / 256)), dim3(256), 0, s,
                           static_cast<const __bf16*>(Y[m]), base + l.y8, base + l.ysc, nblocks);
        HUPR_LAUNCH(hupr_k_quant_vt_mx8, dim3(N / 64, Bn), dim3(256), 0, s, static_cast<const __bf16*>(V[m]), base + l.vt8,
                           base + l.vsc, N);
    }
    HUPR_LAUNCH_OK("hupr_k_quant_mx8");
    return HUPR_OK;
}

// step 2: one attention of the level on the quantised operands: keys = projection kslot of map kmap, queries = projection qslot of
// map qmap, values = map vmap.  Vres: fp32 value map for the residual form (exact add) or null; out fp32 (Bn, N, C); lse (Bn, N);
// out16 (optional): bf16 copy with ld16 elements between tokens
extern "C" int hupr_attn_mx8_fwd(const void* ws, int kmap, int kslot, int qmap, int qslot, int vmap, const float* Vres, float* out,
                                 float* lse, void* out16, int ld16, int Bn, int N, int C, size_t ws_bytes, hupr_stream_t stream) {
    if (int rc = mx8_check("hupr_attn_mx8_fwd", Bn, N, C, ws, ws_bytes)) return rc;
    HUPR_REQUIRE(out && lse && (unsigned)kmap < 2 && (unsigned)qmap < 2 && (unsigned)vmap < 2 && (unsigned)kslot < 4 && (unsigned)qslot < 4,
                 "hupr_attn_mx8_fwd: bad argument");
    HUPR_REQUIRE(!out16 || ld16 % 4 == 0, "hupr_attn_mx8_fwd: ld16 must be a multiple of 4");
    const Mx8Layout l = mx8_layout(Bn, N, C);
    const unsigned char* base = static_cast<const unsigned char*>(ws);
    const unsigned char* kb = base + kmap * l.per_map;
    const unsigned char* qb = base + qmap * l.per_map;
    const unsigned char* vb = base + vmap * l.per_map;
    HUPR_LAUNCH(hupr_k_attn_fwd_mx8, dim3(N / 128, Bn), dim3(256), 0, as_stream(stream), kb + l.y8 + kslot * C, 4 * C,
                       kb + l.ysc + kslot * (C / 32), 4 * C / 32, qb + l.y8 + qslot * C, 4 * C, qb + l.ysc + qslot * (C / 32), 4 * C / 32,
                       vb + l.vt8, vb + l.vsc, Vres, out, lse, static_cast<__bf16*>(out16), ld16, N);
    HUPR_LAUNCH_OK("hupr_k_attn_fwd_mx8");
    return HUPR_OK;
}
