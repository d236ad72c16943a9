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
};

constexpr int kWgHaloVox = 2 * 10 * 10;      // td plane of the 2x8x8 tile; the 1x8x16 tile needs 1*10*18 = 180
constexpr int kRowB = 128;                   // bytes per LDS row: 64 bf16 channels

__device__ __forceinline__ bf16x8 tr_pair(const __bf16* base, int off0, int off1) {
    // two transpose-reads (4 rows each) -> 8 consecutive K values for this lane's column
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(reinterpret_cast<const char*>(base) + off0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(reinterpret_cast<const char*>(base) + off1));
    union { struct { s16x4 a, b; } s; bf16x8 v; } u;
    u.s.a = lo;
    u.s.b = hi;
    return u.v;
}

template <bool ABF>
__global__ __launch_bounds__(256) void hupr_k_wgrad_halo_bf16(WgradHaloArgs p) {
    __shared__ __attribute__((aligned(16))) __bf16 Xh[kWgHaloVox * 64];
    __shared__ __attribute__((aligned(16))) __bf16 DYs[128 * 64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;        // 32-row (co) / 32-col (ci) quadrant of the 64x64 tile
    const int TW = 1 << p.log2TW, TD = p.TD;
    const int HH = 10, HW = TW + 2;
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

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nvox_h = TD * HH * HW;
    for (int st = group; st < p.n_spatial; st += p.groups) {
        int q = st;
        const int twi = q % p.nw; q /= p.nw;
        const int thi = q % p.nh; q /= p.nh;
        const int tdi = q % p.nd;
        const int b = q / p.nd;
        const int d0 = tdi * TD, h0 = thi * 8, w0 = twi * TW;

        __syncthreads();                               // previous tile's fragments are consumed
        // ---- stage x halo (td plane) and dy tile, fp32 -> bf16, batched loads -----------------------------
        const int items_x = nvox_h * 8, items = items_x + 128 * 8;
        constexpr int NB = ABF ? 6 : 4;                // 16-byte (bf16) / 32-byte (fp32) items in flight per thread
        for (int it0 = tid; it0 < items; it0 += NB * 256) {
            float4 va[ABF ? 1 : NB], vc[ABF ? 1 : NB];
            u32x4 vb[ABF ? NB : 1];
            int dst[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int it = it0 + u * 256;
                if constexpr (ABF) vb[u] = (u32x4){0u, 0u, 0u, 0u};
                else { va[u] = make_float4(0.f, 0.f, 0.f, 0.f); vc[u] = va[u]; }
                dst[u] = -1;
                long off = -1;
                bool is_x = true;
                if (it < items_x) {
                    const int vox = it >> 3, c8 = it & 7;
                    const int hx = vox % HW;
                    const int t2 = vox / HW;
                    const int hy = t2 % HH, hz = t2 / HH;
                    const int d = d0 + hz + td - pd, h = h0 + hy - 1, w = w0 + hx - 1;
                    dst[u] = vox * 64 + c8 * 8;
                    if ((unsigned)d < (unsigned)p.D && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W &&
                        ci0 + c8 * 8 < p.Ci)       // Cin = 32 layers: the upper half of the 64-wide tile is zero
                        off = ((((long)b * p.D + d) * p.H + h) * p.W + w) * p.in_ld + ci0 + c8 * 8;
                } else if (it < items) {
                    const int j = it - items_x;
                    const int v = j >> 3, c8 = j & 7;
                    const int wx = v & (TW - 1), hy = (v >> p.log2TW) & 7, dz = v >> (p.log2TW + 3);
                    dst[u] = 0x40000000 | (v * 64 + c8 * 8);
                    is_x = false;
                    if (co0 + c8 * 8 < p.Co)
                        off = ((((long)b * p.D + d0 + dz) * p.H + h0 + hy) * p.W + w0 + wx) * p.dy_ld + co0 + c8 * 8;
                }
                if (off >= 0) {
                    if constexpr (ABF) {
                        vb[u] = *reinterpret_cast<const u32x4*>(static_cast<const __bf16*>(is_x ? p.x : p.dy) + off);
                    } else {
                        const float* src = static_cast<const float*>(is_x ? p.x : p.dy) + off;
                        va[u] = *reinterpret_cast<const float4*>(src);
                        vc[u] = *reinterpret_cast<const float4*>(src + 4);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                if (dst[u] >= 0) {
                    __bf16* base = (dst[u] & 0x40000000) ? DYs : Xh;
                    if constexpr (ABF) {
                        *reinterpret_cast<u32x4*>(&base[dst[u] & 0x3fffffff]) = vb[u];
                    } else {
                        bf16x8 v;
                        v[0] = (__bf16)va[u].x; v[1] = (__bf16)va[u].y; v[2] = (__bf16)va[u].z; v[3] = (__bf16)va[u].w;
                        v[4] = (__bf16)vc[u].x; v[5] = (__bf16)vc[u].y; v[6] = (__bf16)vc[u].z; v[7] = (__bf16)vc[u].w;
                        *reinterpret_cast<bf16x8*>(&base[dst[u] & 0x3fffffff]) = v;
                    }
                }
            }
        }
        __syncthreads();

        // ---- 8 K-steps of 16 voxels; the dy fragment is shared by the 9 taps -------------------------------
#pragma unroll 1
        for (int ks = 0; ks < 8; ++ks) {
            // rows supplied by this lane for the two transpose-reads: tile voxels v = 16 ks + 8 kh + 4 t + rsub
            int vrow[2], xrow[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int v = 16 * ks + 8 * kh_ + 4 * t + rsub;
                const int wx = v & (TW - 1), hy = (v >> p.log2TW) & 7, dz = v >> (p.log2TW + 3);
                vrow[t] = v * kRowB;
                xrow[t] = ((dz * HH + hy) * HW + wx) * kRowB;
            }
            const bf16x8 a = tr_pair(DYs, vrow[0] + wm * 64 + colb, vrow[1] + wm * 64 + colb);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int toff = ((tap / 3) * HW + (tap % 3)) * kRowB;
                const bf16x8 bq = tr_pair(Xh, xrow[0] + toff + wn * 64 + colb, xrow[1] + toff + wn * 64 + colb);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bq, acc[tap], 0, 0, 0);
            }
        }
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

}  // namespace hupr

using namespace hupr;

extern "C" size_t hupr_conv3x3_wgrad_halo_ws_bytes(int Ci, int Co, int kd) {
    // at most 256 groups, but never more than 128 MiB of partials
    const size_t one = (size_t)Co * kd * 9 * Ci * sizeof(float);
    size_t groups = 128;
    while (groups > 1 && groups * one > ((size_t)128 << 20)) groups >>= 1;
    return groups * one;
}

static int wgrad_halo(const void* x, const void* dy, float* dw, int Bn, int D, int H, int W, int Ci, int in_ld, int Co,
                      int dy_ld, int kd, void* ws, size_t ws_bytes, bool abf, hupr_stream_t stream, const char* who) {
    HUPR_REQUIRE(x && dy && dw && ws, "%s: null pointer", who);
    const int al = abf ? 8 : 4;
    HUPR_REQUIRE(Bn > 0 && Co > 0 && Ci % 8 == 0 && Co % 8 == 0 && in_ld % al == 0 && dy_ld % al == 0,
                 "%s: unsupported channels Ci=%d Co=%d", who, Ci, Co);
    HUPR_REQUIRE(H % 8 == 0 && ((kd == 3 && D % 2 == 0 && W % 8 == 0) || (kd == 1 && D == 1 && W % 16 == 0)),
                 "%s: unsupported geometry", who);
    WgradHaloArgs a;
    a.x = x; a.dy = dy; a.part = reinterpret_cast<float*>(ws);
    a.Bn = Bn; a.D = D; a.H = H; a.W = W; a.Ci = Ci; a.in_ld = in_ld; a.Co = Co; a.dy_ld = dy_ld;
    a.kd = kd;
    if (kd == 3) { a.TD = 2; a.log2TW = 3; } else { a.TD = 1; a.log2TW = 4; }
    a.nd = D / a.TD; a.nh = H / 8; a.nw = W >> a.log2TW;
    a.n_ci_tiles = (Ci + 63) / 64;
    a.n_co_tiles = (Co + 63) / 64;
    a.n_spatial = Bn * a.nd * a.nh * a.nw;
    const int pairs = a.n_ci_tiles * a.n_co_tiles * kd;
    const size_t one = (size_t)Co * kd * 9 * Ci * sizeof(float);
    int groups = max(1, min(128, 768 / pairs));      // ~3 workgroups per CU, at most 128 partial tensors
    groups = min(groups, a.n_spatial);
    while (groups > 1 && (size_t)groups * one > ws_bytes) groups >>= 1;
    if ((size_t)groups * one > ws_bytes) return fail(HUPR_ERR_WORKSPACE, "hupr_conv3x3_wgrad_halo_bf16: workspace too small");
    a.groups = groups;
    hipStream_t s = as_stream(stream);
    const dim3 grid(groups, kd, a.n_ci_tiles * a.n_co_tiles);
    if (abf) hipLaunchKernelGGL(hupr_k_wgrad_halo_bf16<true>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(hupr_k_wgrad_halo_bf16<false>, grid, dim3(256), 0, s, a);
    HUPR_LAUNCH_OK("hupr_k_wgrad_halo_bf16");
    const long n = (long)Co * kd * 9 * Ci;
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
extern "C" int hupr_conv3x3_wgrad_halo_bf16act(const void* x, const void* dy, float* dw, int Bn, int D, int H, int W,
                                               int Ci, int in_ld, int Co, int dy_ld, int kd, void* ws, size_t ws_bytes,
                                               hupr_stream_t stream) {
    return wgrad_halo(x, dy, dw, Bn, D, H, W, Ci, in_ld, Co, dy_ld, kd, ws, ws_bytes, true, stream,
                      "hupr_conv3x3_wgrad_halo_bf16act");
}
