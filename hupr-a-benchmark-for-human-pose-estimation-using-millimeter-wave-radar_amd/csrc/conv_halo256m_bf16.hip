// The 256-voxel persistent halo convolution (3 x 3 (x 3) taps, bf16-stored activations) on v_mfma_f32_16x16x32_bf16 — the kernel
// the encoder's and decoder's convolutions and input gradients run on (conv_halo_bf16.hip holds the 128-voxel kernel for everything
// outside this envelope: fp32-stored activations, small grids).  Since round 6 the first layer's two odd shapes run here as well: Ci = 32
// forward (KC = 32) and the 64 -> 32 input gradient (NWN = 1).
//
// Design, as the measurements of rounds 2-5 shaped it (64 -> 64 channels, 3 x 3 x 3 taps, B = 32):
//   * one persistent 512-thread workgroup per CU walks a contiguous run of 4 x 8 x 8-voxel x 64-channel tiles (an eighth of the
//     tile sequence per XCD); the NEXT tile's halo travels global -> registers one item per tap underneath the current tile's
//     MFMAs and is committed to LDS behind the last stage's barrier (a fill burst of all CUs otherwise saturates L2 for ~3 us per
//     tile while the matrix pipe idles);
//   * weights arrive by LDS-DMA (buffer_load ... lds) in stages of three taps = one (kz, kx) column, double-buffered; the stage
//     barrier sits IN FRONT of a stage's last tap and the fragment pipeline runs across stage and tile boundaries: a wave reaches
//     the barrier holding the fragments of the last tap and leaves it with eight MFMAs ready (matrix pipe busy 58 -> 65 % of the
//     cycles, profiles/r04b_conv_sq_pmc.txt);
//   * the instruction shape: at the chip's power limit the 16 x 16 x 32 form does 15-17 % more work per joule than 32 x 32 x 16
//     (register-fed; 5-7 % fed from LDS at this kernel's rate: scripts/probes/mfma_shapes_probe.hip, profiles/r04b_halo_ablation_m16.txt);
//   * a finished tile is parked as packed bf16 and stored under the next tile's first stage; a residual is prefetched under the
//     tile's last stage; BatchNorm column sums of the rounded outputs are kept in registers and reduced once per launch.
// Who owns what:
//   wave = depth slice wm (4) x 32-channel half wn (2); lane = (idx = lane & 15, kq = lane >> 4);
//   the wave's 64 voxels are four groups vg of 16 (rows 2 vg, 2 vg + 1; idx = 8 yy + wx), its 32 channels two groups cg of 16;
//   D'[channel][voxel] = W X^T: lane holds voxel idx of every group and channels 16 cg + 4 kq .. + 3: c[vg][cg] (eight f32x4);
//   a K-step is 32 input channels (lane kq reads the 16-byte chunk 4 kk + kq of a row); per tap ky: 2 weight fragments (cg) and the
//   4 activation fragments of halo rows rho = 2 vg + ky feed 8 MFMAs; rho = 2, 4, 6 serve ky = 0 and ky = 2 (9 activation reads per
//   K-step for 12 (vg, ky) pairs); two register banks of four activation fragments rotate (the even-rho bank of K-step kk + 1 is
//   loaded into the odd-rho bank of kk while ky = 2 multiplies): 13 fragments live, 52 registers.
#include "conv_halo.h"

namespace hupr {

typedef float f32x4c __attribute__((ext_vector_type(4)));

// Tile 4 x 8 x 8 (TD = 4, TW = 8: the encoder levels with D % 4 == 0) or 2 x 8 x 16 (TD = 2, TW = 16: level 3, D = 2 — before this
// variant those layers ran on the 128-voxel kernel at 935 TF/s); a wave owns depth slice wm (TD = 4) or (depth slice wm >> 1, column
// half wm & 1) (TD = 2): 8 x 8 voxels either way.
// 1 x 16 x 16 with KD = 1 (the decoder's 1 x 3 x 3 convolutions on 64 x 64 and 32 x 32 maps, before on the 128-voxel kernel at
// 590-940 TF/s): three stages of three taps, a wave owns one 8 x 8 quadrant.
// NCOT > 0: fused BatchNorm statistics; NCOT = how many DISTINCT 64-wide output-channel tiles one workgroup's run of consecutive
// tiles touches = min(tiles per workgroup, output tiles of the layer): 1 or 2 (round 5: registers instead of the LDS slots of round 4,
// which existed for layers of ONE output tile on the 4 x 8 x 8 kernel only — encoder level 1; levels 2 and 3, Co = 128 / 256, ran a
// statistics pass over the stored tensor per BatchNorm: 24 launches per step).  Level 2 at the bench batch: 4 tiles per workgroup,
// alternating between its 2 output tiles; level 3: one tile per workgroup.
// NWN = 1 (round 6; Co = 32: the input gradient of the encoders' first convolution, 64 -> 32 channels, before on the 128-voxel kernel
// at 702 TF/s): the eight waves are eight 8 x 8 voxel blocks of an 8 x 8 x 8 tile (512 voxels), each with all 32 output channels —
// per wave the same 8 MFMAs per tap from the same fragment reads as the 64-channel form; a weight stage is 3 x 4 KB (two DMA
// instructions, the second by waves 0-3 only); sixteen halo items per thread ride under the first sixteen taps.
// KC = 32 (round 6; Ci = 32: the encoders' first convolution, in rounds 2-5 on a 512-voxel register-blocked kernel of its own — 118-127 us
// against 115 here, profiles/r06_conv_ci32_ab.txt; removed): a halo / weight row is 64 bytes,
// a tap is ONE K-step, a stage is a kz plane with its nine (kx, ky) taps (nine 4 KB weight images by 36 LDS-DMA pieces); the
// 64-byte rows have their own bank keys (see the fragment addresses).  The kernel walks Ci / 32 chunks per tile, so it also takes 64-channel
// layers (two chunks: the two halo images then replace the one 64-channel image); measured on layer 1 (64 -> 64, B = 32): 192 us against
// 182 us of the KC = 64 form (207 us with one image and a commit burst per chunk) — not dispatched (profiles/r06_conv_ci32_ab.txt).
template <int TD, int TH, int TW, int KD, int NCOT = 0, int NWN = 2, int KC = 64>
__global__ __launch_bounds__(512) void hupr_k_conv_halo256m_bf16(HaloArgs p) {
    constexpr int LDK = KC, BN = 32 * NWN, TS = KC == 64 ? 3 : 9, C8 = KC / 8, L2C8 = KC == 64 ? 3 : 2;
    static_assert(KC == 64 || (KC == 32 && KD == 3 && NWN == 2 && NCOT == 0), "32-channel rows: 3 x 3 x 3 taps, 64 output channels per tile");
    constexpr int HD = TD + KD - 1, HH = TH + 2, HW = TW + 2;
    static_assert(TD * TH * TW == 64 * (8 / NWN) && TH % 8 == 0 && TW % 8 == 0 && (KD == 1 || KD == 3) && (NWN == 2 || NCOT == 0),
                  "256 (512) voxels per tile, 8 x 8 per wave");
    constexpr int NVOX = HD * HH * HW;                         // 600 (4 x 8 x 8) / 720 (2 x 8 x 16) halo voxels
    constexpr int T = 9 * KD, NSTAGE = KC == 64 ? 3 * KD : KD, NTAP = KC == 64 ? 6 : 9;      // stage = (kz, kx); its taps: K-step kk (2) x ky (3)  [KC = 32: stage = kz; taps: kx (3) x ky (3)]
    // NWN = 1: the tile spans the whole depth (launcher: D == TD), so the halo's first and last planes are the zero padding — never
    // loaded, zeroed once in the prologue; the items cover planes 1 .. TD only (13 per thread instead of 16: the register file is full)
    constexpr bool FULLD = NWN == 1;
    constexpr int NVOXL = FULLD ? TD * HH * HW : NVOX, VOX0 = FULLD ? HH * HW : 0;
    constexpr int NH = (NVOXL * C8 + 511) / 512;               // 10 / 12 / 13 halo items (8 channels of a voxel) per thread
    // items are issued one per tap from the item's first tap on; in front of the barriers of stages 0, 1, 2 (each in front of the stage's
    // sixth tap) the items of taps 6 s - 1 .. 6 s + 4 are younger than the weight pieces the barrier waits for
    // (stage s: the items NTAP s - 1 .. NTAP s + NTAP - 2)
    constexpr int Y0 = NH < NTAP - 1 ? NH : NTAP - 1, Y1 = NH - (NTAP - 1) < 0 ? 0 : (NH - (NTAP - 1) > NTAP ? NTAP : NH - (NTAP - 1)),
                  Y2 = NH - (2 * NTAP - 1) < 0 ? 0 : (NH - (2 * NTAP - 1) > NTAP ? NTAP : NH - (2 * NTAP - 1));
    static_assert(NH <= NTAP * NSTAGE - 1 && NH <= 3 * NTAP - 1, "halo items must all be issued in front of the item's last barrier (and within three stages)");
    // KC = 32: TWO halo images (2 x 38 KB — what the one 64-channel image takes): item q multiplies image q & 1 while the halo of item q + 1
    // is committed to the other one, one ds_write_b128 per tap under the MFMAs of the item's last stage — no commit burst and no barrier
    // at the item boundary (the last stage's barrier already stands between those writes and the first reads of the next item)
    __shared__ __attribute__((aligned(16))) __bf16 Hs[(KC == 32 ? 2 : 1) * NVOX * LDK];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[2][TS][BN * LDK];
    // fused BatchNorm statistics: per lane and output-channel tile the running sums of its eight channels over its voxels (bf16-ROUNDED
    // outputs) in REGISTERS — ssum / ssq [tile][cg][r] — reduced over the sixteen voxel lanes and the four voxel-block waves once,
    // after the tile loop, in double (through the halo image's LDS, dead by then)
    constexpr bool STATS = NCOT > 0 && KD == 3;
    constexpr bool SINGLE = STATS && TD == 2;                  // the 2 x 8 x 16 tile carries statistics only for ONE tile per workgroup
    f32x4c ssum[STATS ? NCOT : 1][2], ssq[STATS ? NCOT : 1][2];
    if constexpr (!SINGLE) {                                   // (SINGLE: assigned once, in the tile's epilogue — not live across the tap loop)
#pragma unroll
        for (int ct = 0; ct < (STATS ? NCOT : 1); ++ct)
#pragma unroll
            for (int cg = 0; cg < 2; ++cg) { ssum[ct][cg] = (f32x4c){0.f, 0.f, 0.f, 0.f}; ssq[ct][cg] = (f32x4c){0.f, 0.f, 0.f, 0.f}; }
    }

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    const int wm = NWN == 2 ? wave >> 1 : wave, wn = NWN == 2 ? wave & 1 : 0;
    constexpr int NBX = TW / 8, NBY = TH / 8;
    const int xw0 = 8 * (wm % NBX), yw0 = 8 * ((wm / NBX) % NBY), dzw = wm / (NBX * NBY);      // the wave's 8 x 8 block: first column, first row, depth slice
    const int idx = lane & 15, kq = lane >> 4, yy = idx >> 3, wx = idx & 7;
    const int n_tiles = p.Bn * p.nd * p.nh * p.nw * p.n_co_tiles;

    // ---- weight stages by LDS-DMA ---------------------------------------------------------------------------------------
    // (NWN = 1: a DMA instruction moves TWO taps of 32 rows — waves 0-3 the first, waves 4-7 the second)
    const int wrow_ = NWN == 2 ? 8 * wave + (lane >> 3) : ((8 * wave + (lane >> 3)) & 31);
    const int wsrc_lane = (wrow_ * T * p.Ci + (((lane & 7) ^ (wrow_ >> 1)) & 7) * 8) * 2 +
                          (NWN == 2 ? 0 : (wave >> 2) * (3 * p.Ci * 2));                       // bytes; < 2^31
    constexpr int NDMA = NWN == 2 ? TS : 2, DMA_TAPS = NWN == 2 ? 1 : 2;
    // KC = 32: piece pc = 8 j + wave (1 KB = 16 rows of 64 B) of a stage's 36: tap slot pc >> 2 = kx * 3 + ky, rows 16 (pc & 3) + (lane >> 2);
    // chunk c of weight row n lives at position c ^ (((n >> 3) & 1) << 1) (the four rows n = v mod 4 of a ds_read_b128 lane group then
    // hold four different positions)
    const int wrow32_ = 16 * (wave & 3) + (lane >> 2);
    const int wsrc32_lane = (wrow32_ * T * p.Ci + (((lane & 3) ^ (((wrow32_ >> 3) & 1) << 1)) & 3) * 8) * 2;
    const u32x4 wrs = {(unsigned)(unsigned long)p.wp, (unsigned)((unsigned long)p.wp >> 32) & 0xffffu,
                       (unsigned)((long)p.Co * T * p.Ci * 2), 0x00020000u};
    const unsigned bs_lds = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)&Bs[0][0][0];
    const unsigned wdst_wave = __builtin_amdgcn_readfirstlane(bs_lds + wave * 1024);
#define HUPR_W_DMA(COT_, CH_, S_, PAR_)                                                                             \
    if constexpr (KC == 32) {                                                                                       \
        _Pragma("unroll") for (int j = 0; j < 5; ++j) {                                                             \
            if (j < 4 || wave < 4) {                                                                                \
                const int t9_ = 2 * j + (wave >> 2);             /* wave-uniform: kx * 3 + ky */                     \
                const int wb32_ = (((COT_) * BN * T + (S_) * 9 + (t9_ % 3) * 3 + t9_ / 3) * p.Ci + (CH_) * KC) * 2 + wsrc32_lane; \
                unsigned keep_;                                                                                     \
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\t" \
                             "s_mov_b32 m0, %0"                                                                     \
                             : "=&s"(keep_)                                                                         \
                             : "s"(wdst_wave + (PAR_) * TS * (BN * LDK * 2) + j * 8192), "v"(wb32_), "s"(wrs)       \
                             : "memory");                                                                           \
            }                                                                                                       \
        }                                                                                                           \
    } else {                                                                                                        \
        const int wbase_ = (((COT_) * BN * T + ((S_) / 3) * 9 + ((S_) % 3)) * p.Ci + (CH_) * KC) * 2 + wsrc_lane;   \
        _Pragma("unroll") for (int j = 0; j < NDMA; ++j) {                                                          \
            if (NWN == 2 || j == 0 || wave < 4) {                                                                   \
                unsigned keep_;                                                                                     \
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\t" \
                             "s_mov_b32 m0, %0"                                                                     \
                             : "=&s"(keep_)                                                                         \
                             : "s"(NWN == 2 ? wdst_wave + ((PAR_) * TS + j) * (BN * LDK * 2)                        \
                                            : wdst_wave + (PAR_) * TS * (BN * LDK * 2) + j * 8192),                 \
                               "v"(NWN == 2 ? wbase_ + j * (3 * p.Ci * 2) : wbase_ + j * DMA_TAPS * (3 * p.Ci * 2)), "s"(wrs) \
                             : "memory");                                                                           \
            }                                                                                                       \
        }                                                                                                           \
    }

    // ---- halo: global -> registers -> LDS ------------------------------------------------------------------------------
    u32x4 vb[NH];
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.x), 0, (int)((long)p.Bn * p.D * p.H * p.W * p.in_ld * 2), 0x00020000);
#define HUPR_HALO_ISSUE_ITEM(u, COND_, B_, D0_, H0_, W0_, C0_)                                                     \
    {                                                                                                               \
        const int it = tid + (u) * 512;                                                                             \
        const int vox = it >> L2C8, c8 = it & (C8 - 1);                                                             \
        const int hx = vox % HW;                                                                                    \
        const int t_ = vox / HW;                                                                                    \
        const int hy = t_ % HH, hz = t_ / HH;                                                                       \
        const int d = (D0_) + hz - (FULLD ? 0 : KD / 2), h = (H0_) + hy - 1, w = (W0_) + hx - 1;                    \
        const bool ok = (COND_) && it < NVOXL * C8 && (unsigned)d < (unsigned)p.D && (unsigned)h < (unsigned)p.H && \
                        (unsigned)w < (unsigned)p.W;                                                                \
        const int off = (((((B_) * p.D + d) * p.H + h) * p.W + w) * p.in_ld + (C0_) + c8 * 8) * 2;                  \
        const auto ld_ = __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? off : 0x7ffffff0, 0, 0);                  \
        vb[u] = (u32x4){ld_[0], ld_[1], ld_[2], ld_[3]};                                                            \
    }
#define HUPR_HALO_COMMIT()                                                                                          \
    _Pragma("unroll") for (int u = 0; u < NH; ++u) {                                                                \
        const int it = tid + u * 512;                                                                               \
        if (it < NVOXL * C8) {                                                                                      \
            const int vox = it >> L2C8, c8 = it & (C8 - 1);                                                         \
            const int hx = vox % HW;                                                                                \
            const int key_ = KC == 64 ? (((hx >> 1) & 3) << 1) : ((((vox / HW) % HH) & 1) << 1);                    \
            *reinterpret_cast<u32x4*>(&Hs[(vox + VOX0) * LDK + ((c8 ^ key_) << 3)]) = vb[u];                        \
        }                                                                                                           \
    }

    struct Pos { int cot, twi, thi, tdi, b, ch; };
    const int n_chunks = p.Ci / KC;
    const int per_wg = (n_tiles + gridDim.x - 1) / gridDim.x;
    const int wg_rank = (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));      // an eighth of the tile sequence per XCD
    const int t_begin = wg_rank * per_wg, t_end = min(n_tiles, t_begin + per_wg);
    if (STATS && p.stats) {
        if (t_begin >= t_end) {
            for (int c = tid; c < 2 * p.Co; c += 512) p.stats[(long)blockIdx.x * 2 * p.Co + c] = 0.0;
        }
    }
    if (t_begin >= t_end) return;
    Pos cur;
    {
        cur.cot = t_begin % p.n_co_tiles;
        int st = t_begin / p.n_co_tiles;
        cur.twi = st % p.nw; st /= p.nw;
        cur.thi = st % p.nh; st /= p.nh;
        cur.tdi = st % p.nd;
        cur.b = st / p.nd;
        cur.ch = 0;
    }
    const int n_items = (t_end - t_begin) * n_chunks;
    const int cot0 = cur.cot;                                  // first output tile of this workgroup's run (statistics)

    // ---- fragment addresses ----------------------------------------------------------------------------------------
    // weights: row n = 32 wn + 16 cg + idx of a tap's [64][64] image, chunk 4 kk + kq at position chunk ^ ((n >> 1) & 7)
    int woff[2][2];
#pragma unroll
    for (int cg = 0; cg < 2; ++cg)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int n = 32 * wn + 16 * cg + idx;
            woff[cg][kk] = KC == 64 ? n * LDK + (((4 * kk + kq) ^ ((n >> 1) & 7)) << 3) : n * LDK + ((kq ^ (((n >> 3) & 1) << 1)) << 3);
        }
    // activations: halo voxel (wm + kz, rho + yy, wx + kx).  Halo row swizzle of THIS kernel: 16-byte chunk c of voxel (hy, hx) lives at
    // chunk c ^ (((hx >> 1) & 3) << 1).  A 16-lane ds_read_b128 group holds eight lanes of chunk parity 0 and eight of parity 1 such
    // that the two rows yy of a column differ in that parity; voxel pitch 128 B puts the column parity into bank bit 5; the key separates
    // the four columns of one parity in chunk bits 1-2: all 64 banks, every tap (SQ_LDS_BANK_CONFLICT = 0, profiles/r04b_conv_sq_pmc.txt)
    const int xlane = ((dzw * HH + yw0 + yy) * HW + xw0 + wx) * LDK;
    // KC = 32 (64-byte rows): a voxel's four chunks sit at chunk ^ ((halo row & 1) << 1).  A ds_read_b128 lane group holds, for each
    // voxel class (index mod 4 = a 64-byte bank quarter), the four lanes (kq, row), (kq, row + 1), (kq + 1, row), (kq + 1, row + 1): positions
    // kq ^ {0, 2} and (kq + 1) ^ {0, 2} — all four.  KK_ there = kx (the stage is the kz plane ST_).
#define HUPR_HALO_COMMIT1(u, HW_)                                                                                   \
    {                                                                                                               \
        const int it = tid + (u) * 512;                                                                             \
        if (it < NVOXL * C8) {                                                                                      \
            const int vox = it >> L2C8, c8 = it & (C8 - 1);                                                         \
            const int key_ = (((vox / HW) % HH) & 1) << 1;                                                          \
            *reinterpret_cast<u32x4*>(&Hs[(HW_) + (vox + VOX0) * LDK + ((c8 ^ key_) << 3)]) = vb[u];                \
        }                                                                                                           \
    }
#define HUPR_XF(ST_, RHO_, KK_)                                                                                     \
    (KC == 64 ? *reinterpret_cast<const bf16x8*>(&Hs[xlane + ((((ST_) / 3) * HH + (RHO_)) * HW + ((ST_) % 3)) * LDK + \
                    (((4 * (KK_) + kq) ^ ((((xw0 + wx + ((ST_) % 3)) >> 1) & 3) << 1)) << 3)])                      \
              : *reinterpret_cast<const bf16x8*>(&Hs[hrd + xlane + (((ST_) * HH + (RHO_)) * HW + (KK_)) * LDK +     \
                    ((kq ^ (((yy + (RHO_)) & 1) << 1)) << 3)]))
#define HUPR_WF(BUF_, KY_, CG_, KK_)                                                                                \
    (KC == 64 ? *reinterpret_cast<const bf16x8*>(&Bs[BUF_][KY_][woff[CG_][KK_]])                                    \
              : *reinterpret_cast<const bf16x8*>(&Bs[BUF_][3 * (KK_) + (KY_)][woff[CG_][0]]))

    f32x4c c[4][2];
    unsigned accP[4][2][2];                                      // parked tile: bf16 pairs of c[vg][cg]
    __bf16* pP = nullptr;                                        // parked tile: this lane's 16-byte run of voxel group 0
    bool pend = false;
    // A residual no longer forces the immediate epilogue (round 5; the second input gradient of a block's convolution pair, the decoder's
    // conv + residual: 228 us against 184 us at level 1): this lane's 32 residual elements are fetched into registers under the MFMAs
    // of the tile's LAST stage — their loads have returned by that stage's vmcnt(0) barrier — added in fp32 before the one rounding,
    // and the tile is parked and stored under the next tile's first stage like any other.  (Not in the instantiations that have no
    // 16 registers to spare: fused statistics, which never carry a residual, and the 2 x 8 x 16 tile.)
    constexpr bool RESPF = !STATS && TD != 2 && NWN == 2;
    const bool res_pf = RESPF && p.res != nullptr && (p.res_ld & 3) == 0 && !p.no_res_prefetch;
    // (KC = 32 — the encoders' first convolution carries a bias —: the lane's eight bias values are fetched at the top of a tile and added in
    // front of the one rounding; the other instantiations have no registers for them and their layers no bias)
    constexpr bool BIASPF = KC == 32;
    const bool defer = (!p.bias || BIASPF) && (!p.res || res_pf) && (p.Co & 7) == 0 && (p.out_ld & 7) == 0;
    float biasv[BIASPF ? 2 : 1][4];
    bf16x4 resv[4][2];
    // (KC = 32: THREE fragment banks — a stage has an odd number of tap groups (kx = 0, 1, 2) and of taps (9), so with two banks the first
    // fragments of the next stage would land in the bank the stage's last tap is still multiplying: even rows of group kx in bank kx, odd
    // rows in bank (kx + 1) % 3, weights of tap tau in set tau % 3.  KC = 64: two K-steps, six taps: banks kk & 1, sets tau & 1 as always.)
    constexpr int NB = KC == 64 ? 2 : 3;
    bf16x8 xq[NB][4], x8, wq[NB][2];
    // fragments of tap TAU_ (= 3 kk + ky) of stage ST_ from weight buffer BUF_ (WX_: 1 weights only, 2 activations only, 3 both)
#define HUPR_LOAD_TAP(ST_, TAU_, BUF_, WX_)                                                                         \
    {                                                                                                               \
        constexpr int kk_ = (TAU_) / 3, ky_ = (TAU_) % 3;                                                           \
        if ((WX_) & 1) {                                                                                            \
            wq[(TAU_) % NB][0] = HUPR_WF(BUF_, ky_, 0, kk_);                                                        \
            wq[(TAU_) % NB][1] = HUPR_WF(BUF_, ky_, 1, kk_);                                                        \
        }                                                                                                           \
        if ((WX_) & 2) {                                                                                            \
            if (ky_ == 0) { _Pragma("unroll") for (int vg = 0; vg < 4; ++vg) xq[kk_ % NB][vg] = HUPR_XF(ST_, 2 * vg, kk_); } \
            else if (ky_ == 1) { _Pragma("unroll") for (int vg = 0; vg < 4; ++vg) xq[(kk_ + 1) % NB][vg] = HUPR_XF(ST_, 2 * vg + 1, kk_); } \
            else { x8 = HUPR_XF(ST_, 8, kk_); }                                                                     \
        }                                                                                                           \
    }
#define HUPR_VMCNT_LGKM0(N_) __builtin_amdgcn_s_waitcnt(0x0070 | ((N_) & 15) | (((N_) >> 4) << 14))

    // prologue: first item's halo, weight stages 0 and 1
    if constexpr (FULLD) {
        for (int i = tid; i < 2 * HH * HW * C8; i += 512) {
            const int pl = i / (HH * HW * C8), r = i - pl * (HH * HW * C8);
            *reinterpret_cast<u32x4*>(&Hs[pl * (HD - 1) * HH * HW * LDK + r * 8]) = (u32x4){0u, 0u, 0u, 0u};
        }
    }
    HUPR_W_DMA(cur.cot, cur.ch, 0, 0)
    HUPR_W_DMA(cur.cot, cur.ch, 1, 1)
#pragma unroll
    for (int u = 0; u < NH; ++u) HUPR_HALO_ISSUE_ITEM(u, true, cur.b, cur.tdi * TD, cur.thi * TH, cur.twi * TW, cur.ch * KC)
    [[maybe_unused]] int hrd = 0;                                 // (KC = 32) element offset of the halo image this item reads
    if constexpr (KC == 32) {
#pragma unroll
        for (int u = 0; u < NH; ++u) HUPR_HALO_COMMIT1(u, 0)
    } else {
        HUPR_HALO_COMMIT()
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    HUPR_LOAD_TAP(0, 0, 0, 3)

    int g = 0;                                                    // global stage counter: stage g reads Bs[g & 1]
    for (int q = 0; q < n_items; ++q) {
        const int b = cur.b, d0 = cur.tdi * TD, h0 = cur.thi * TH, w0 = cur.twi * TW, n0 = cur.cot * BN;
        const bool first_chunk = cur.ch == 0, last_chunk = cur.ch == n_chunks - 1;
        Pos nxt = cur;
        if (++nxt.ch == n_chunks) {
            nxt.ch = 0;
            if (++nxt.cot == p.n_co_tiles) {
                nxt.cot = 0;
                if (++nxt.twi == p.nw) {
                    nxt.twi = 0;
                    if (++nxt.thi == p.nh) {
                        nxt.thi = 0;
                        if (++nxt.tdi == p.nd) { nxt.tdi = 0; ++nxt.b; }
                    }
                }
            }
        }
        const bool has_next = q + 1 < n_items;
        if constexpr (KC == 32) hrd = (q & 1) * (NVOX * LDK);
        [[maybe_unused]] const int hwr = NVOX * LDK - hrd;            // (KC = 32) ... and the one the next item's halo is written to
        if (first_chunk) {
#pragma unroll
            for (int vg = 0; vg < 4; ++vg)
#pragma unroll
                for (int cg = 0; cg < 2; ++cg) c[vg][cg] = (f32x4c){0.f, 0.f, 0.f, 0.f};
            if constexpr (BIASPF) {
                if (p.bias && defer) {
#pragma unroll
                    for (int cg = 0; cg < 2; ++cg)
#pragma unroll
                        for (int r = 0; r < 4; ++r) biasv[cg][r] = p.bias[n0 + 32 * wn + 16 * cg + 4 * kq + r];
                }
            }
        }
#pragma unroll
        for (int st_ = 0; st_ < NSTAGE; ++st_) {
            const int par = (g + st_) & 1;
#pragma unroll
            for (int tau = 0; tau < NTAP; ++tau) {
                const int kk = tau / 3, ky = tau % 3;
                if (tau == NTAP - 1) {
                    // the stage's barrier: this wave's reads of Bs[par] have all returned and its pieces of stage s + 1 have landed in
                    // Bs[par ^ 1]; younger operations stay in flight: the next halo's register loads issued since (one per tap: five
                    // in front of the first two barriers of an item) and in stage 0 the four stores of the parked tile
                    if (st_ == 0) { if (pend) { HUPR_VMCNT_LGKM0(Y0 + 4); } else { HUPR_VMCNT_LGKM0(Y0); } }
                    else if (st_ == 1) { HUPR_VMCNT_LGKM0(Y1); }
                    else if (st_ == 2) { HUPR_VMCNT_LGKM0(Y2); }
                    else { HUPR_VMCNT_LGKM0(0); }
                    __syncthreads();
                    if (st_ + 2 < NSTAGE) { HUPR_W_DMA(cur.cot, cur.ch, st_ + 2, par) }
                    else if (has_next) { HUPR_W_DMA(nxt.cot, nxt.ch, st_ + 2 - NSTAGE, par) }
                    if (KC == 64 && st_ == NSTAGE - 1 && has_next) { HUPR_HALO_COMMIT() }
                }
                // the next tap's fragments: of this stage, or tap 0 of the next one (across an item boundary only its weights)
                if (tau == 0) { HUPR_LOAD_TAP(st_, 1, par, 3) }
                else if (tau == 1) { HUPR_LOAD_TAP(st_, 2, par, 3) }
                else if (tau == 2) { HUPR_LOAD_TAP(st_, 3, par, 3) }
                else if (tau == 3) { HUPR_LOAD_TAP(st_, 4, par, 3) }
                else if (tau == 4) { HUPR_LOAD_TAP(st_, 5, par, 3) }
                else if (NTAP == 9 && tau == 5) { if constexpr (NTAP == 9) { HUPR_LOAD_TAP(st_, 6, par, 3) } }
                else if (NTAP == 9 && tau == 6) { if constexpr (NTAP == 9) { HUPR_LOAD_TAP(st_, 7, par, 3) } }
                else if (NTAP == 9 && tau == 7) { if constexpr (NTAP == 9) { HUPR_LOAD_TAP(st_, 8, par, 3) } }
                else if (st_ + 1 < NSTAGE) { HUPR_LOAD_TAP((st_ + 1) % NSTAGE, 0, par ^ 1, 3) }
                else if (has_next) {
                    if constexpr (KC == 32) { const int hrd = hwr; HUPR_LOAD_TAP(0, 0, par ^ 1, 3) }      // (the next image is complete: this stage's barrier)
                    else { HUPR_LOAD_TAP(0, 0, par ^ 1, 1) }
                }
                // eight MFMAs: activation rows rho = 2 vg + ky
#pragma unroll
                for (int vg = 0; vg < 4; ++vg) {
                    const bf16x8 xf = ky == 0 ? xq[kk % NB][vg] : (ky == 1 ? xq[(kk + 1) % NB][vg] : (vg < 3 ? xq[kk % NB][vg + 1] : x8));
                    c[vg][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[tau % NB][0], xf, c[vg][0], 0, 0, 0);
                    c[vg][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[tau % NB][1], xf, c[vg][1], 0, 0, 0);
                }
                if (st_ == 0 && pend && tau < 4) {
                    // one voxel group of the parked tile per tap.  v_permlane16_swap trades the odd 16-lane rows of the cg = 0 dwords for the
                    // even rows of the cg = 1 dwords: a kq-even lane then holds channels 4 kq .. 4 kq + 7 of group 0, a kq-odd lane
                    // channels 4 (kq - 1) .. + 7 of group 1 — ONE 16-byte store per lane and voxel instead of two 8-byte ones
                    const auto r0 = __builtin_amdgcn_permlane16_swap(accP[tau][0][0], accP[tau][1][0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane16_swap(accP[tau][0][1], accP[tau][1][1], false, false);
                    *reinterpret_cast<u32x4*>(pP + (long)(2 * tau * p.W) * p.out_ld) = (u32x4){r0[0], r1[0], r0[1], r1[1]};
                }
                if (RESPF && st_ == NSTAGE - 1 && tau == 0) {
                    if (res_pf && last_chunk && defer) {
                        const long mr = (((long)b * p.D + d0 + dzw) * p.H + h0 + yw0 + yy) * p.W + w0 + xw0 + wx;
                        const int chr = n0 + 32 * wn + 4 * kq;
#pragma unroll
                        for (int vg = 0; vg < 4; ++vg)
#pragma unroll
                            for (int cg = 0; cg < 2; ++cg)
                                resv[vg][cg] = *reinterpret_cast<const bf16x4*>(static_cast<const __bf16*>(p.res) + (mr + 2 * vg * p.W) * p.res_ld +
                                                                                chr + 16 * cg);
                    }
                }
                if constexpr (KC == 32) {                         // the next item's halo -> the idle image, one item per tap of the last stage
                    if (st_ == NSTAGE - 1 && tau < NH && has_next) { HUPR_HALO_COMMIT1(tau, hwr) }
                }
                // the next item's halo: one item per tap under its MFMAs (branch-free: an out-of-range offset past the last item)
                if (NTAP * st_ + tau < NH) {
                    HUPR_HALO_ISSUE_ITEM(NTAP * st_ + tau, has_next, nxt.b, nxt.tdi * TD, nxt.thi * TH, nxt.twi * TW, nxt.ch * KC)
                }
                // a fragment read in front of the MFMAs, one at a time (six reads at most, eight MFMAs)
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
                for (int i_ = 0; i_ < 4; ++i_) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (st_ == 0) pend = false;
        }
        g += NSTAGE;
        if (last_chunk) {
            // lane's voxel of group vg: (dz = wm, row 2 vg + yy, column wx); its channels: n0 + 32 wn + 16 cg + 4 kq .. + 3
            const long m0 = (((long)b * p.D + d0 + dzw) * p.H + h0 + yw0 + yy) * p.W + w0 + xw0 + wx;
            const int ch0 = n0 + 32 * wn + 4 * kq;
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            if (defer && has_next) {                              // park: stored during the next item's stage 0
                if constexpr (BIASPF) {
                    if (p.bias) {
#pragma unroll
                        for (int vg = 0; vg < 4; ++vg)
#pragma unroll
                            for (int cg = 0; cg < 2; ++cg)
#pragma unroll
                                for (int r = 0; r < 4; ++r) c[vg][cg][r] += biasv[cg][r];
                    }
                }
                if (RESPF && res_pf) {
#pragma unroll
                    for (int vg = 0; vg < 4; ++vg)
#pragma unroll
                        for (int cg = 0; cg < 2; ++cg)
#pragma unroll
                            for (int r = 0; r < 4; ++r) c[vg][cg][r] += (float)resv[vg][cg][r];
                }
#pragma unroll
                for (int vg = 0; vg < 4; ++vg)
#pragma unroll
                    for (int cg = 0; cg < 2; ++cg) {
                        const bf16x2 lo = {(__bf16)c[vg][cg][0], (__bf16)c[vg][cg][1]}, hi = {(__bf16)c[vg][cg][2], (__bf16)c[vg][cg][3]};
                        accP[vg][cg][0] = __builtin_bit_cast(unsigned, lo);
                        accP[vg][cg][1] = __builtin_bit_cast(unsigned, hi);
                    }
                pP = static_cast<__bf16*>(p.y) + m0 * p.out_ld + n0 + 32 * wn + ((kq & 1) ? 16 + 4 * (kq - 1) : 4 * kq);
                pend = true;
            } else {
#pragma unroll
                for (int vg = 0; vg < 4; ++vg)
#pragma unroll
                    for (int cg = 0; cg < 2; ++cg) {
                        const long m = m0 + 2 * vg * p.W;
                        const int ch = ch0 + 16 * cg;
                        if (ch < p.Co) {
                            f32x4n v = {c[vg][cg][0], c[vg][cg][1], c[vg][cg][2], c[vg][cg][3]};
                            if (p.bias) v += (f32x4n){p.bias[ch], p.bias[ch + 1], p.bias[ch + 2], p.bias[ch + 3]};
                            if (p.res) {
                                const bf16x4 rv = *reinterpret_cast<const bf16x4*>(static_cast<const __bf16*>(p.res) + m * p.res_ld + ch);
                                v += (f32x4n){(float)rv[0], (float)rv[1], (float)rv[2], (float)rv[3]};
                            }
                            const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                            *reinterpret_cast<bf16x4*>(static_cast<__bf16*>(p.y) + m * p.out_ld + ch) = o;
                        }
                    }
            }
            if (STATS && p.stats) {
                // this lane's eight channels over its four voxels of the tile, rounded exactly as they are stored
                int li = cur.cot - cot0;                           // which of the workgroup's (at most NCOT) output tiles: workgroup-uniform
                li += li < 0 ? p.n_co_tiles : 0;
                if constexpr (SINGLE) {
                    // one tile per workgroup (the launcher guarantees it): the sums START here — nothing is carried through the tap loop
#pragma unroll
                    for (int cg = 0; cg < 2; ++cg) {
                        f32x4c sv = {0.f, 0.f, 0.f, 0.f}, qv = sv;
#pragma unroll
                        for (int vg = 0; vg < 4; ++vg)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float a = (float)(__bf16)c[vg][cg][r];
                                sv[r] += a;
                                qv[r] = fmaf(a, a, qv[r]);
                            }
                        ssum[0][cg] = sv;
                        ssq[0][cg] = qv;
                    }
                } else {
#pragma unroll
                    for (int ct = 0; ct < NCOT; ++ct)
                        if (NCOT == 1 || li == ct) {
#pragma unroll
                            for (int cg = 0; cg < 2; ++cg)
#pragma unroll
                                for (int vg = 0; vg < 4; ++vg)
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        const float a = (float)(__bf16)c[vg][cg][r];
                                        ssum[ct][cg][r] += a;
                                        ssq[ct][cg][r] = fmaf(a, a, ssq[ct][cg][r]);
                                    }
                        }
                }
            }
        }
        if (KC == 64 && has_next) {
            __syncthreads();                                      // the halo committed behind the last stage's barrier is complete
            HUPR_LOAD_TAP(0, 0, g & 1, 2)                         // its weight fragments were read under the last MFMAs above
        }
        cur = nxt;
    }
    if (STATS && p.stats) {
        // channel ch = 32 wn + 16 cg + 4 kq + r of tile ct collects, in a fixed order and as doubles, the sixteen voxel lanes of its (kq)
        // row group in each of the four voxel-block waves; tiles this workgroup never multiplied contribute their zeros
        static_assert(NVOX * LDK * 2 >= 8 * 64 * 16 * 4, "the halo image must hold the reduction scratch");
        float (*St)[64][16] = reinterpret_cast<float (*)[64][16]>(Hs);
#pragma unroll
        for (int ct = 0; ct < NCOT; ++ct) {
            __syncthreads();
#pragma unroll
            for (int cg = 0; cg < 2; ++cg) {
                *reinterpret_cast<f32x4c*>(&St[wave][lane][4 * cg]) = ssum[ct][cg];
                *reinterpret_cast<f32x4c*>(&St[wave][lane][8 + 4 * cg]) = ssq[ct][cg];
            }
            __syncthreads();
            if (tid < 2 * BN) {
                const int k = tid >> 6, ch = tid & 63;
                const int wn_ = ch >> 5, cg = (ch >> 4) & 1, kq_ = (ch >> 2) & 3, r = ch & 3;
                double t = 0.0;
#pragma unroll
                for (int wm_ = 0; wm_ < 4; ++wm_)
#pragma unroll
                    for (int i = 0; i < 16; ++i) t += (double)St[2 * wm_ + wn_][16 * kq_ + i][8 * k + 4 * cg + r];
                p.stats[(long)blockIdx.x * 2 * p.Co + k * p.Co + 64 * ((cot0 + ct) % p.n_co_tiles) + ch] = t;
            }
        }
        if (tid < 2 * BN) {                                       // output tiles this workgroup never touched: zeros (same writer per entry)
            const int k = tid >> 6, ch = tid & 63;
            for (int ct = NCOT; ct < p.n_co_tiles; ++ct)
                p.stats[(long)blockIdx.x * 2 * p.Co + k * p.Co + 64 * ((cot0 + ct) % p.n_co_tiles) + ch] = 0.0;
        }
    }
#undef HUPR_W_DMA
#undef HUPR_HALO_ISSUE_ITEM
#undef HUPR_HALO_COMMIT
#undef HUPR_HALO_COMMIT1
#undef HUPR_XF
#undef HUPR_WF
#undef HUPR_LOAD_TAP
#undef HUPR_VMCNT_LGKM0
}

static void launch_conv_halo256m(const HaloArgs& a, hipStream_t s) {
    const dim3 grid(kHalo256Grid), wg(512);
    if (a.Ci == 32) HUPR_LAUNCH((hupr_k_conv_halo256m_bf16<4, 8, 8, 3, 0, 2, 32>), grid, wg, 0, s, a);
    else if (a.TD == 8) HUPR_LAUNCH((hupr_k_conv_halo256m_bf16<8, 8, 8, 3, 0, 1>), grid, wg, 0, s, a);
    else if (a.kd == 1) HUPR_LAUNCH((hupr_k_conv_halo256m_bf16<1, 16, 16, 1>), grid, wg, 0, s, a);
    else if (a.stats) {                       // fused BatchNorm statistics: 1 or 2 distinct output tiles per workgroup (conv_halo256_stats_ok)
        const long tiles = (long)a.Bn * a.nd * a.nh * a.nw * a.n_co_tiles;
        const long per_wg = (tiles + kHalo256Grid - 1) / kHalo256Grid;
        const bool one = per_wg == 1 || a.n_co_tiles == 1;
        if (a.TD == 4) {
            if (one) HUPR_LAUNCH((hupr_k_conv_halo256m_bf16<4, 8, 8, 3, 1>), grid, wg, 0, s, a);
            else HUPR_LAUNCH((hupr_k_conv_halo256m_bf16<4, 8, 8, 3, 2>), grid, wg, 0, s, a);
        } else {
            HUPR_LAUNCH((hupr_k_conv_halo256m_bf16<2, 8, 16, 3, 1>), grid, wg, 0, s, a);      // per_wg == 1 (conv_halo256_stats_ok)
        }
    } else if (a.TD == 4) HUPR_LAUNCH((hupr_k_conv_halo256m_bf16<4, 8, 8, 3>), grid, wg, 0, s, a);
    else HUPR_LAUNCH((hupr_k_conv_halo256m_bf16<2, 8, 16, 3>), grid, wg, 0, s, a);
}

// ---- which launches the 256-voxel kernel takes (bf16-stored activations only; everything else: the 128-voxel kernel) ----------
// Test aid (hupr_debug_halo_tiles): bit 0 = the 4 x 8 x 8 tile, bit 1 = the 2 x 8 x 16 tile (depth not a multiple of four: encoder
// level 3), bit 2 = the 1 x 16 x 16 tile (1 x 3 x 3 taps: the decoder).  A cleared bit sends those layers to the 128-voxel kernel —
// the comparison the parity tests make (same products, another fp32 summation order).  Bit 3 = the 8 x 8 x 8 tile of the 32-output-
// channel form, bit 4 = the 32-input-channel form.
static int g_halo_tiles = 31;
void set_halo_tiles(int mask) { g_halo_tiles = mask & 31; }

static bool offsets_fit(const HaloArgs& a, int Bn) { return (long)Bn * a.D * a.H * a.W * a.in_ld * 2 < 0x7ffffff0L; }      // 32-bit buffer offsets

bool conv_halo256_supported(const HaloArgs& a, int Bn, bool abf) {
    if (!abf || !(g_halo_tiles & 1)) return false;
    if (a.kd != 3 || a.D % 4 != 0 || a.H % 8 != 0 || a.W % 8 != 0 || a.Ci % 64 != 0 || a.Co % 64 != 0) return false;
    const long tiles = (long)Bn * (a.D / 4) * (a.H / 8) * (a.W / 8) * (a.Co / 64);
    if (tiles >= (1L << 31) || tiles < 256) return false;      // small problems: the 128-voxel kernel fills the chip better
    return offsets_fit(a, Bn);
}

// Would a launch with fused BatchNorm statistics (a.stats) land on an instantiation that has them?  Co = 64: the 4 x 8 x 8 tile's
// one-tile form; Co = 128 / 256 (2 / 4 output tiles): register-resident sums of at most two distinct output tiles per workgroup;
// the 2 x 8 x 16 tile: exactly one tile per workgroup (encoder level 3 at B = 32).
bool conv_halo256_stats_ok(const HaloArgs& a, int Bn) {
    if (a.kd != 3 || a.Ci % 64 != 0 || a.Co % 64 != 0 || !offsets_fit(a, Bn)) return false;
    const int n = a.Co / 64;
    if (a.D % 4 == 0) {
        if (!conv_halo256_supported(a, Bn, true)) return false;
        if (n == 1) return true;
        const long tiles = (long)Bn * (a.D / 4) * (a.H / 8) * (a.W / 8) * n;
        const long per_wg = (tiles + kHalo256Grid - 1) / kHalo256Grid;
        return n <= 2 || per_wg == 1;
    }
    if (!(g_halo_tiles & 2) || a.D % 2 != 0 || a.H % 8 != 0 || a.W % 16 != 0) return false;
    return (long)Bn * (a.D / 2) * (a.H / 8) * (a.W / 16) * n == 256;
}

bool launch_conv_halo256(HaloArgs a, int Bn, bool abf, hipStream_t s) {
    if (!abf || !offsets_fit(a, Bn)) return false;
    if (a.Ci == 32) {
        // 32 input channels (the encoders' first convolution): the 4 x 8 x 8 tile on 64-byte rows, one K-step per tap
        if (!(g_halo_tiles & 16) || a.kd != 3 || a.D % 4 != 0 || a.H % 8 != 0 || a.W % 8 != 0 || a.Co % 64 != 0 || a.stats ||
            (a.in_ld & 7) || (long)a.Co * 27 * a.Ci * 2 >= 0x7ffffff0L)
            return false;
        a.n_co_tiles = a.Co / 64;
        a.TD = 4; a.log2TW = 3;
        a.nd = a.D / 4; a.nh = a.H / 8; a.nw = a.W / 8;
        const long tiles_ = (long)Bn * a.nd * a.nh * a.nw * a.n_co_tiles;
        if (tiles_ < 256 || tiles_ >= (1L << 31)) return false;
        launch_conv_halo256m(a, s);
        return true;
    }
    if (a.Ci % 64 != 0) return false;
    if (a.Co == 32) {
        // 32 output channels (the first layer's input gradient): the 8 x 8 x 8 tile, every wave with all the channels
        if (!(g_halo_tiles & 8) || a.kd != 3 || a.D != 8 || a.H % 8 != 0 || a.W % 8 != 0 || a.stats || (a.out_ld & 7)) return false;
        a.n_co_tiles = 1;
        a.TD = 8; a.log2TW = 3;
        a.nd = a.D / 8; a.nh = a.H / 8; a.nw = a.W / 8;
        const long tiles32 = (long)Bn * a.nd * a.nh * a.nw;
        if (tiles32 < 256 || tiles32 >= (1L << 31)) return false;
        launch_conv_halo256m(a, s);
        return true;
    }
    if (a.Co % 64 != 0) return false;
    a.n_co_tiles = a.Co / 64;
    if (a.kd == 3 && a.D % 4 == 0) {
        if (!conv_halo256_supported(a, Bn, abf)) return false;
        a.TD = 4; a.log2TW = 3;
        a.nd = a.D / 4; a.nh = a.H / 8; a.nw = a.W / 8;
    } else if (a.kd == 3) {
        // depth not a multiple of four (encoder level 3: D = 2): the 2 x 8 x 16 tile
        if (!(g_halo_tiles & 2) || a.D % 2 != 0 || a.H % 8 != 0 || a.W % 16 != 0) return false;
        a.TD = 2; a.log2TW = 4;
        a.nd = a.D / 2; a.nh = a.H / 8; a.nw = a.W / 16;
    } else {
        // 1 x 3 x 3 convolutions of the decoder: the 1 x 16 x 16 tile (no fused statistics)
        if (!(g_halo_tiles & 4) || a.kd != 1 || a.D != 1 || a.H % 16 != 0 || a.W % 16 != 0 || a.stats) return false;
        a.TD = 1; a.log2TW = 4;
        a.nd = 1; a.nh = a.H / 16; a.nw = a.W / 16;
    }
    const long tiles = (long)Bn * a.nd * a.nh * a.nw * a.n_co_tiles;
    if (tiles < 256 || tiles >= (1L << 31)) return false;
    launch_conv_halo256m(a, s);      // one persistent workgroup per CU
    return true;
}

}  // namespace hupr
