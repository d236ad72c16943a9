// 256-voxel persistent variant of the halo-tiled 3x3x3 convolution (bf16 matrix pipe) for the 3-D encoder layers.
//
// What the measurements said (scripts/halo_ablation.py, scripts/halo_trace.py; 64->64 3x3x3 at B = 32):
//   * the 128-voxel kernel re-streams 221 KB of weights per 128 voxels and pays fill / store / barrier phases per
//     small tile -> a 512-thread workgroup (8 waves) owns a 4x8x8 tile, one persistent workgroup per CU walks a
//     contiguous range of tiles, and the NEXT tile's halo travels global -> registers underneath the current tile's
//     MFMA stages (the fill burst of all CUs otherwise saturates L2 for ~3 us per tile while the matrix pipe idles);
//   * inside the tap loop the kernel was bound by LDS READ bandwidth, not by the matrix pipe: 1.5 fragment reads
//     (1.5 KB) per 32x32x16 MFMA is 192 B/clk/CU against 128 B/clk.  A lane's two output rows are therefore
//     neighbours in y (hy, hy + 1) and a stage walks the three ky taps of one (kz, kx) column: the four halo rows
//     hy .. hy + 3 are read once and feed all six (row, ky) products — 7 reads per 6 MFMAs instead of 9.
// Requirements: kd = 3, Ci % 64 == 0, Co % 64 == 0, D % 4 == 0, H % 8 == 0, W % 8 == 0, >= 256 tiles; otherwise the
// 128-voxel kernel (conv_halo_bf16.hip) is used.
#include "conv_halo.h"

namespace hupr {

// ABL: compile-time phase ablation for scripts/halo_ablation.py (wrong results, timing only): 1 = no weight staging
// (stages read whatever Bs holds), 2 = no per-stage barrier, 4 = fragments of K-step 0 only (no re-reads inside a stage)
//      16 = the rounds-1-4 stage protocol (barrier behind a stage's last K-step, fragment pipeline restarted after it), A/B aid
template <bool ABF, int ABL = 0>
__global__ __launch_bounds__(512) void hupr_k_conv_halo256_bf16(HaloArgs p) {
    constexpr int KC = 64, LDK = 64, BN = 64, TS = 3, C8 = 8;
    // PL (bf16 activations, no ablation): the stage barrier sits IN FRONT of a stage's last K-step instead of behind it, and the
    // fragment pipeline runs across stage boundaries.  What the SQ counters said about the old protocol
    // (profiles/r04_conv_sq_pmc.txt): matrix pipe busy 58 % of the cycles, the waves parked at a wait or the barrier 36 % of
    // theirs — behind every barrier all eight waves asked for the same 56 KB of first fragments at once and the pipe idled until
    // they arrived.  Now a wave reaches the barrier holding the fragments of the stage's last K-step (all its reads of the
    // stage's weight buffer have returned, so the barrier also frees that buffer: the LDS-DMA of stage s + 2 is issued right
    // behind it), leaves it with six MFMAs ready, and reads the first fragments of stage s + 1 — whose weights every wave waited
    // for before the barrier — under them.  Same two buffers, same MFMA order, same bits.
    constexpr bool PL = ABF && (ABL == 0 || ABL == 32);
    // ABL 32 (timing only, wrong results): every 32x32x16 MFMA of the tap loop replaced by two v_mfma_f32_16x16x32_bf16 on the same
    // fragment registers and eight 16 x 16 accumulators — same bytes, same flops, the other instruction shape (DESIGN.md section 4,
    // "Which matrix-instruction shape does the most work per joule"): what a conversion of this kernel would buy
    constexpr bool M16 = ABL == 32;
    typedef float f32x4t __attribute__((ext_vector_type(4)));
    f32x4t acc16[M16 ? 8 : 1];
#pragma unroll
    for (int i = 0; i < (M16 ? 8 : 1); ++i) acc16[i] = (f32x4t){0.f, 0.f, 0.f, 0.f};
    constexpr int TD = 4, TH = 8, TW = 8, HD = TD + 2, HH = TH + 2, HW = TW + 2;
    constexpr int NVOX = HD * HH * HW;                         // 600 halo voxels
    constexpr int T = 27, NSTAGE = 9;                          // stage = (kz, kx), its three taps = ky 0..2
    constexpr int NI = (NVOX * C8 + 511) / 512;                // 10 halo items (8 channels of a voxel) per thread ...
    constexpr int NH = ABF ? NI : NI / 2;                      // ... fp32 sources: two half batches (register budget)
        __shared__ __attribute__((aligned(16))) __bf16 Hs[NVOX * LDK];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[2][TS][BN * LDK];     // double-buffered weight stages
    // fused BatchNorm statistics (p.stats; Co == 64, one co tile): running sums of the bf16-ROUNDED outputs per lane PAIR —
    // [wave][lane >> 1][value r = 0..15 sums, 16..31 sums of squares].  A tile costs one DPP pair-add per value and a plain
    // read-add-write of the pair's own 128-byte slot (16-byte LDS accesses; nobody else touches it: no atomics — ds_add_f32
    // retires about one lane per clock and cost 80 us per launch); the cross-lane / cross-wave reduction happens ONCE after
    // the tile loop, and the deferred epilogue stays on.  (Rounds 1-2 reduced all 32 voxel lanes
    // per tile: 160 DPP adds + 32 LDS adds, immediate epilogue — 72 % of the cost of the statistics pass it replaced.  Per-lane
    // register accumulators would be cheaper still, but the kernel sits at 245 of its 256 VGPRs.)
    __shared__ __attribute__((aligned(16))) float St[8][32][36];      // [wave][lane >> 1][32 values + 4 pad: conflict-free 16-byte accesses]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // static issue priority for the second-dispatched half of the workgroup (it loses every arbitration against its older
    // SIMD partner otherwise): -2% kernel time measured on the layer-1 shape, either half works, no per-segment flips
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    const int wm = wave >> 1, wn = wave & 1;                   // wave = depth slice dz (4) x 32-channel half (2)
    const int lr = lane & 31, lh = lane >> 5;
    // MFMA column lr of accumulator tile i  <->  output voxel (dz = wm, hy = 2 (lr >> 3) + i, wx = lr & 7)
    const int wx = lr & 7, hy0 = 2 * (lr >> 3);
    const int abase = ((wm * HH + hy0) * HW + wx) * LDK;       // halo element of tap (0,0,0) of row i = 0
    const int ekey = ((lr >> 3) & 1) << 2;                     // swizzle bit 2 of halo rows hy0, hy0+1 (flipped for +2, +3)
    const int bkey = ((wn * 32 + lr) >> 1) & 7;
    const int n_tiles = p.Bn * p.nd * p.nh * p.nw * p.n_co_tiles;

    // Weight stages travel global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no staging VGPRs, no ds_write burst behind
    // every barrier; measured -19 us of the 191 us tap loop on the layer-1 shape).  A stage (3 taps x 64 rows x 128 B) is
    // 24 pieces of 1 KiB = 8 rows; wave w moves rows 8 w .. 8 w + 7 of each tap: lane l deposits 16 bytes at piece base +
    // 16 l = row 8 w + (l >> 3), chunk position l & 7.  The row swizzle (chunk c8 of row n lives at position c8 ^ (n >> 1))
    // is applied on the SOURCE side: position c' is filled with source chunk c' ^ ((n >> 1) & 7).
    // The DMA is issued from inline asm (M0 = LDS address, saved / restored around it): hipcc then neither counts it nor
    // guards later LDS reads with vmcnt(0); the stage protocol below does its own counted waits.
    const int wrow_ = 8 * wave + (lane >> 3);
    const int wsrc_lane = (wrow_ * T * p.Ci + (((lane & 7) ^ (wrow_ >> 1)) & 7) * 8) * 2;      // bytes; < 2^31
    const u32x4 wrs = {(unsigned)(unsigned long)p.wp, (unsigned)((unsigned long)p.wp >> 32) & 0xffffu,
                       (unsigned)((long)p.Co * T * p.Ci * 2), 0x00020000u};
    const unsigned bs_lds = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)&Bs[0][0][0];
    const unsigned wdst_wave = __builtin_amdgcn_readfirstlane(bs_lds + wave * 1024);

    f32x16 acc[2];
    // deferred epilogue (bf16 output, no bias / residual / fused statistics, Co % 8 == 0): a finished tile's accumulators are
    // parked in accP and stored in four pieces between the MFMA groups of the next item's first stage
    unsigned accP[2][8];                                          // parked tile, already rounded to bf16 pairs
    long mP = 0;
    int chP = 0;
    bool pend = false;
    const bool defer = ABF && !p.bias && !p.res && (p.Co & 7) == 0 && (p.out_ld & 7) == 0 && !(p.ablate & (4 | 2));      // ablate bit 1: A/B switch (immediate epilogue)
    f32x4n va[ABF ? 1 : NH], vc[ABF ? 1 : NH];
    u32x4 vb[ABF ? NH : 1];

    // halo row swizzle: 16-byte chunk c8 of halo voxel (hy, hx) lives at chunk c8 ^ (((hx >> 1) & 3) | (((hy >> 1) & 1) << 2));
    // with the lane -> voxel map above every 16-lane ds_read_b128 group covers all 64 banks for all 27 tap shifts.
    // ISSUE: global -> registers for halo items [U0, U0 + NH) of the tile at (B_, D0_, H0_, W0_), channel chunk C0_.
    // bf16 sources: branch-free raw buffer loads (offsets beyond num_records return zeros = the padding); fp32 sources
    // (the secondary path) keep guarded flat loads
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.x), 0, ABF ? (int)((long)p.Bn * p.D * p.H * p.W * p.in_ld * 2) : 0, 0x00020000);
#define HUPR_HALO_ISSUE_ITEM(u, U0, COND_, B_, D0_, H0_, W0_, C0_)                                                 \
    {                                                                                                               \
        const int it = tid + ((u) + (U0)) * 512;                                                                    \
        const int vox = it >> 3, c8 = it & 7;                                                                       \
        const int hx = vox % HW;                                                                                    \
        const int t_ = vox / HW;                                                                                    \
        const int hy = t_ % HH, hz = t_ / HH;                                                                       \
        const int d = (D0_) + hz - 1, h = (H0_) + hy - 1, w = (W0_) + hx - 1;                                       \
        const bool ok = (COND_) && it < NVOX * C8 && !(p.ablate & 1) && (unsigned)d < (unsigned)p.D &&              \
                        (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;                                 \
        if constexpr (ABF) {                                                                                        \
            const int off = (((((B_) * p.D + d) * p.H + h) * p.W + w) * p.in_ld + (C0_) + c8 * 8) * 2;              \
            const auto ld_ = __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? off : 0x7ffffff0, 0, 0);              \
            vb[u] = (u32x4){ld_[0], ld_[1], ld_[2], ld_[3]};                                                        \
        } else {                                                                                                    \
            va[u] = (f32x4n){0.f, 0.f, 0.f, 0.f};                                                                   \
            vc[u] = va[u];                                                                                          \
            if (ok) {                                                                                               \
                const float* src = static_cast<const float*>(p.x) +                                                 \
                                   ((((long)(B_) * p.D + d) * p.H + h) * p.W + w) * p.in_ld + (C0_) + c8 * 8;       \
                va[u] = *reinterpret_cast<const f32x4n*>(src);                                                      \
                vc[u] = *reinterpret_cast<const f32x4n*>(src + 4);                                                  \
            }                                                                                                       \
        }                                                                                                           \
    }
#define HUPR_HALO_ISSUE(U0, B_, D0_, H0_, W0_, C0_)                                                                \
    _Pragma("unroll") for (int u = 0; u < NH; ++u) HUPR_HALO_ISSUE_ITEM(u, U0, true, B_, D0_, H0_, W0_, C0_)
    // COMMIT: registers -> LDS (zeros outside the tensor)
#define HUPR_HALO_COMMIT(U0)                                                                                       \
    _Pragma("unroll") for (int u = 0; u < NH; ++u) {                                                                \
        const int it = tid + (u + (U0)) * 512;                                                                      \
        if (it < NVOX * C8 && !(p.ablate & 1)) {                                                                    \
            const int vox = it >> 3, c8 = it & 7;                                                                   \
            const int hx = vox % HW, hy = (vox / HW) % HH;                                                          \
            __bf16* dstp = &Hs[vox * LDK + ((c8 ^ (((hx >> 1) & 3) | (((hy >> 1) & 1) << 2))) << 3)];               \
            if constexpr (ABF) {                                                                                    \
                *reinterpret_cast<u32x4*>(dstp) = vb[u];                                                            \
            } else {                                                                                                \
                bf16x8 v;                                                                                           \
                v[0] = (__bf16)va[u].x; v[1] = (__bf16)va[u].y; v[2] = (__bf16)va[u].z; v[3] = (__bf16)va[u].w;     \
                v[4] = (__bf16)vc[u].x; v[5] = (__bf16)vc[u].y; v[6] = (__bf16)vc[u].z; v[7] = (__bf16)vc[u].w;     \
                *reinterpret_cast<bf16x8*>(dstp) = v;                                                               \
            }                                                                                                       \
        }                                                                                                           \
    }
    // profiling: s_memtime stamps from wave 0 of workgroup 0 (all branches below are wave-uniform)
    const bool tracing = p.trace != nullptr && blockIdx.x == 0 && wave == 0;
    int tslot = 0;
#define HUPR_STAMP()                                                                            \
    if (tracing) {                                                                              \
        const unsigned long long t__ = __builtin_amdgcn_s_memtime();                            \
        if (lane == 0 && tslot < 4096) p.trace[tslot] = t__;                                    \
        ++tslot;                                                                                \
    }

    // Work items = (tile, channel chunk), a contiguous range of tiles per workgroup (co tile fastest, then w, h, d,
    // batch); coordinates advance by carries.  Everything is software-pipelined across stages AND items:
    //   * weights: global -> registers two stages ahead, registers -> the other half of the double-buffered Bs one stage
    //     ahead, so a stage needs ONE barrier and never waits on a load it has just issued;
    //   * halo of item q + 1: global -> registers during stage 0 of item q, registers -> LDS after item q's last stage;
    //   * fragments: the LDS reads of K-step ks + 1 are issued before the MFMAs of K-step ks (two register sets).
    struct Pos { int cot, twi, thi, tdi, b, ch; };
    const int n_chunks = p.Ci / KC;
    const int per_wg = (n_tiles + gridDim.x - 1) / gridDim.x;
    // XCD-aware range assignment: workgroup id -> XCD is id % 8 and every XCD has its own L2, so the eight XCDs each walk
    // ONE contiguous eighth of the tile sequence (neighbouring tile rows / depth slabs, whose halos overlap, then hit in
    // the same L2 instead of being fetched from HBM once per XCD)
    const int wg_rank = (p.ablate & 8) ? (int)blockIdx.x : (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));
    const int t_begin = wg_rank * per_wg, t_end = min(n_tiles, t_begin + per_wg);
    if (p.stats) {
        for (int i = tid; i < 8 * 32 * 36; i += 512) (&St[0][0][0])[i] = 0.f;      // published by the prologue barrier
        if (t_begin >= t_end) {
            for (int c = tid; c < 2 * p.Co; c += 512) p.stats[(long)blockIdx.x * 2 * p.Co + c] = 0.0;
        }
    }
    if (t_begin >= t_end) return;
    if (p.trace != nullptr && tid == 0) p.trace[4096 + 2 * blockIdx.x] = wall_clock64();      // profiling: per-workgroup start / end
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
    // weights of stage S_ (0..8) of the item at co tile COT_ / chunk CH_ -> Bs[PAR_]: taps (kz, ky = 0..2, kx)
#define HUPR_W_DMA(COT_, CH_, S_, PAR_)                                                                             \
    {                                                                                                               \
        const int wbase_ = (((COT_) * BN * T + ((S_) / 3) * 9 + ((S_) % 3)) * p.Ci + (CH_) * KC) * 2 + wsrc_lane;   \
        _Pragma("unroll") for (int j = 0; j < TS; ++j) {                                                            \
            unsigned keep_;                                                                                         \
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\t" \
                         "s_mov_b32 m0, %0"                                                                         \
                         : "=&s"(keep_)                                                                             \
                         : "s"(wdst_wave + ((PAR_) * TS + j) * (BN * LDK * 2)), "v"(wbase_ + j * (3 * p.Ci * 2)), "s"(wrs) \
                         : "memory");                                                                               \
        }                                                                                                           \
    }
    // all but the youngest N_ vector-memory operations of this wave have completed (vmcnt is 6 bits: [3:0] and [15:14])
#define HUPR_VMCNT(N_) __builtin_amdgcn_s_waitcnt(0x0F70 | ((N_) & 15) | (((N_) >> 4) << 14))

    // prologue: first item's halo, weight stage 0 -> Bs[0] (PL: and stage 1 -> Bs[1])
    HUPR_W_DMA(cur.cot, cur.ch, 0, 0)
    if constexpr (PL) { HUPR_W_DMA(cur.cot, cur.ch, 1, 1) }
    HUPR_HALO_ISSUE(0, cur.b, cur.tdi * TD, cur.thi * TH, cur.twi * TW, cur.ch * KC)
    HUPR_HALO_COMMIT(0)
    if constexpr (!ABF) {
        HUPR_HALO_ISSUE(NH, cur.b, cur.tdi * TD, cur.thi * TH, cur.twi * TW, cur.ch * KC)
        HUPR_HALO_COMMIT(NH)
    }
    HUPR_VMCNT(0);
    __syncthreads();

    // fragments of stage ST_, K-step KS_: activations from the halo, weights from buffer BUF_
    bf16x8 af[2][4], bq[2][TS];                                   // two fragment sets: K-step ks + 1 is read while ks multiplies
#define HUPR_FRAGS_A(SET_, ST_, KS_)                                                                                \
    {                                                                                                               \
        const int toff_ = (((ST_) / 3) * HH * HW + ((ST_) % 3)) * LDK;                                              \
        const int xkey_ = ((wx + ((ST_) % 3)) >> 1) & 3;                                                            \
        const int cw_ = (KS_) * 2 + lh;                                                                             \
        _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                               \
            af[SET_][r] = *reinterpret_cast<const bf16x8*>(                                                         \
                &Hs[abase + toff_ + r * (HW * LDK) + ((cw_ ^ (xkey_ | (ekey ^ ((r >> 1) << 2)))) << 3)]);           \
    }
#define HUPR_FRAGS_B(SET_, KS_, BUF_)                                                                               \
    {                                                                                                               \
        const int cw_ = (KS_) * 2 + lh;                                                                             \
        _Pragma("unroll") for (int t = 0; t < TS; ++t)                                                              \
            bq[SET_][t] = *reinterpret_cast<const bf16x8*>(                                                         \
                &Bs[BUF_][t][(wn * 32 + lr) * LDK + ((cw_ ^ bkey) << 3)]);                                          \
    }
    if constexpr (PL) {
        HUPR_FRAGS_A(0, 0, 0)
        HUPR_FRAGS_B(0, 0, 0)
    }
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
        HUPR_STAMP()                                              // 0: item start
        if (first_chunk && !M16) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        }
#pragma unroll
        for (int st_ = 0; st_ < NSTAGE; ++st_) {
            const int par = (g + st_) & 1;
            // weights of the next stage -> the idle half of Bs (everyone left it at the previous barrier).  (A ring of three
            // with the DMA two stages ahead measured no faster: the wait below is not where the staging cost sits.)
            if constexpr (!(ABL & 1) && !PL) {
                if (st_ + 1 < NSTAGE) { HUPR_W_DMA(cur.cot, cur.ch, st_ + 1, par ^ 1) }
                else if (has_next) { HUPR_W_DMA(nxt.cot, nxt.ch, 0, par ^ 1) }
            }
            if (!PL && st_ == 0 && has_next) {                    // next item's halo rides under the remaining stages
                HUPR_HALO_ISSUE(0, nxt.b, nxt.tdi * TD, nxt.thi * TH, nxt.twi * TW, nxt.ch * KC)
            }
            if (st_ == 0) { HUPR_STAMP() }                        // 1: stage-0 issue work done
            if constexpr (PL) {
#pragma unroll
                for (int ks = 0; ks < KC / 16; ++ks) {
                    if (ks == KC / 16 - 1) {
                        // the stage's barrier: this wave's reads of Bs[par] have all returned (lgkmcnt 0) and its pieces of stage
                        // s + 1 have landed in Bs[par ^ 1] (issued behind the previous barrier; younger operations — in stage 0
                        // the next halo's NH register loads and three stores of the parked tile — stay in flight)
                        // (younger than those pieces and left in flight: the next halo's register loads issued since — one
                        // per K-step from the item's first one on, see below — and in stage 0 three stores of the parked tile)
                        // halo items of K-steps 4 s - 1 .. 4 s + 2: three, four, three for s = 0, 1, 2 (NH = 10 items in all)
                        static_assert(NH == 10, "the counted waits below assume ten halo items per thread");
                        if (st_ == 0) {
                            if (pend) { __builtin_amdgcn_s_waitcnt(0x0070 | 6); } else { __builtin_amdgcn_s_waitcnt(0x0070 | 3); }
                        } else if (st_ == 1) {
                            __builtin_amdgcn_s_waitcnt(0x0070 | 4);
                        } else if (st_ == 2) {
                            __builtin_amdgcn_s_waitcnt(0x0070 | 3);
                        } else {
                            __builtin_amdgcn_s_waitcnt(0x0070);
                        }
                        __syncthreads();
                        // Bs[par] is free: weights of stage s + 2
                        if (st_ + 2 < NSTAGE) { HUPR_W_DMA(cur.cot, cur.ch, st_ + 2, par) }
                        else if (has_next) { HUPR_W_DMA(nxt.cot, nxt.ch, st_ + 2 - NSTAGE, par) }
                        // behind the LAST stage's barrier nobody reads this item's halo any more: the next one (all of its register
                        // loads returned by the fourth barrier) goes to LDS under the six MFMAs still to come
                        if (st_ == NSTAGE - 1 && has_next) { HUPR_HALO_COMMIT(0) }
                    }
                    // the next K-step's fragments: of this stage, or K-step 0 of the next one (across an item boundary only its
                    // weights — the halo changes first)
                    if (ks + 1 < KC / 16) {
                        if (ks & 1) { HUPR_FRAGS_A(0, st_, ks + 1) HUPR_FRAGS_B(0, ks + 1, par) }
                        else { HUPR_FRAGS_A(1, st_, ks + 1) HUPR_FRAGS_B(1, ks + 1, par) }
                    } else if (st_ + 1 < NSTAGE) {
                        HUPR_FRAGS_A(0, (st_ + 1) % NSTAGE, 0)
                        HUPR_FRAGS_B(0, 0, par ^ 1)
                    } else if (has_next) {
                        HUPR_FRAGS_B(0, 0, par ^ 1)
                    }
#pragma unroll
                    for (int t = 0; t < TS; ++t) {                // ky;  D'[channel][voxel]
                        if constexpr (M16) {
                            acc16[(4 * t) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[ks & 1][t], af[ks & 1][t], acc16[(4 * t) & 7], 0, 0, 0);
                            acc16[(4 * t + 1) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[ks & 1][t], af[ks & 1][t + 1], acc16[(4 * t + 1) & 7], 0, 0, 0);
                            acc16[(4 * t + 2) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks & 1][t], bq[ks & 1][t], acc16[(4 * t + 2) & 7], 0, 0, 0);
                            acc16[(4 * t + 3) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks & 1][t + 1], bq[ks & 1][t], acc16[(4 * t + 3) & 7], 0, 0, 0);
                        } else {
                            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[ks & 1][t], af[ks & 1][t], acc[0], 0, 0, 0);
                            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[ks & 1][t], af[ks & 1][t + 1], acc[1], 0, 0, 0);
                        }
                    }
                    if (st_ == 0 && pend) {                       // piece ks of the previous tile's epilogue rides under these MFMAs
                        halo_store_packed_part(p, accP[ks >> 1], mP + (ks >> 1) * p.W, chP, (ks & 1) * 2);
                    }
                    // the next item's halo, global -> registers: ONE item (address arithmetic + a 16-byte load) per K-step, under
                    // its MFMAs, from the item's first K-step on (as one block in front of stage 0 it cost 1 850 cycles per tile
                    // during which the wave issued no MFMA: scripts/halo_trace.py, profiles/r04_halo_trace.txt)
                    // (branch-free: past the last item the load is issued with an out-of-range offset — no memory access — so that
                    // the counted waits hold and the scheduler may mix the arithmetic with the MFMAs)
                    if (4 * st_ + ks < NH) {
                        HUPR_HALO_ISSUE_ITEM(4 * st_ + ks, 0, has_next, nxt.b, nxt.tdi * TD, nxt.thi * TH, nxt.twi * TW, nxt.ch * KC)
                    }
#pragma unroll
                    for (int i_ = 0; i_ < 6; ++i_) {              // one fragment read in front of every MFMA (see below)
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, M16 ? 2 : 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                const int kz = st_ / 3, kx = st_ % 3;
                const int toff = (kz * HH * HW + kx) * LDK;
                const int xkey = ((wx + kx) >> 1) & 3;
                const __bf16* Bt = &Bs[par][0][0];
#define HUPR_FRAGS(SET_, KS_)                                                                                       \
                {                                                                                                   \
                    const int cw_ = (KS_) * 2 + lh;                                                                 \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                   \
                        af[SET_][r] = *reinterpret_cast<const bf16x8*>(                                             \
                            &Hs[abase + toff + r * (HW * LDK) + ((cw_ ^ (xkey | (ekey ^ ((r >> 1) << 2)))) << 3)]); \
                    _Pragma("unroll") for (int t = 0; t < TS; ++t)                                                  \
                        bq[SET_][t] = *reinterpret_cast<const bf16x8*>(                                             \
                            &Bt[t * (BN * LDK) + (wn * 32 + lr) * LDK + ((cw_ ^ bkey) << 3)]);                       \
                }
                HUPR_FRAGS(0, 0)
#pragma unroll
                for (int ks = 0; ks < KC / 16; ++ks) {
                    if (ks + 1 < KC / 16 && !(ABL & 4)) {
                        if (ks & 1) { HUPR_FRAGS(0, ks + 1) } else { HUPR_FRAGS(1, ks + 1) }
                    }
                    // keep the machine scheduler from sinking the prefetch reads back next to their uses (it would
                    // shrink the register footprint and re-expose the LDS latency in front of every MFMA)
                    if constexpr (ABL & 8) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < TS; ++t) {                // ky;  D'[channel][voxel]
                        constexpr int fs = (ABL & 4) ? 0 : -1;
                        const int set = fs == 0 ? 0 : (ks & 1);
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[set][t], af[set][t], acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[set][t], af[set][t + 1], acc[1], 0, 0, 0);
                    }
                    if constexpr (ABF) {
                        if (st_ == 0 && pend) {                   // piece ks of the previous tile's epilogue rides under these MFMAs
                            halo_store_packed_part(p, accP[ks >> 1], mP + (ks >> 1) * p.W, chP, (ks & 1) * 2);
                        }
                    }
                    // MFMA / fragment-read order inside a K-step: the seven reads of K-step ks + 1 are independent of the six MFMAs
                    // of ks; issued as one burst (ABL & 8: the round-1 order, sched_barrier between the two groups) the LDS
                    // pipe sees 7 KB per wave at once and then nothing; one read in FRONT of every MFMA measured 3 % faster on
                    // layer 1 (210.5 -> 204.2 us) and 2 % on layer 2 (one read BEHIND every MFMA: 1.7 %; two behind each of the
                    // first three: 1.2 %) — same instructions, same results.
                    if constexpr (!(ABL & 8)) {
#pragma unroll
                        for (int i_ = 0; i_ < 6; ++i_) {
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one DS read
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                        }
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#undef HUPR_FRAGS
            }
            // this wave's pieces of the next stage have landed (in stage 0 the next halo's NH younger register loads stay in
            // flight); after the barrier Bs[par ^ 1] is complete and all waves are done with Bs[par] (and, after stage 8, Hs)
            // (the parked tile's four stores of stage 0 are younger still; guarded fp32 loads have no fixed count)
            if constexpr (!PL) {
                if (ABF && st_ == 0 && has_next) { if (pend) { HUPR_VMCNT(NH + 4); } else { HUPR_VMCNT(NH); } } else { HUPR_VMCNT(0); }
                if constexpr (!(ABL & 2)) __syncthreads();
            }
            if (st_ == 0) { pend = false; HUPR_STAMP() }          // 2: first stage computed (and the parked tile stored)
        }
        g += NSTAGE;
        HUPR_STAMP()                                              // 3: all stages done
        if constexpr (M16) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i & 1][4 * (i >> 1) + j] = acc16[i][j];       // (the large accumulators are dead during the loop)
        }
        if (last_chunk && !(p.ablate & 4)) {
            const long m0 = (((long)b * p.D + d0 + wm) * p.H + h0 + hy0) * p.W + w0 + wx;
            if (defer && has_next) {                              // park: stored during the next item's stage 0
                halo_pack_tile(acc[0], accP[0]);
                halo_pack_tile(acc[1], accP[1]);
                mP = m0;
                chP = n0 + wn * 32 + 4 * lh;
                pend = true;
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) halo_store_voxel<ABF>(p, acc[i], m0 + i * p.W, n0 + wn * 32 + 4 * lh);
            }
        }
        if constexpr (ABF) {
            if (p.stats && last_chunk) {
                // this lane's 16 channels x its two voxels of the tile just finished, rounded exactly as they are stored;
                // + the neighbouring voxel lane (quad_perm [1,0,3,2]); the even lane owns the pair's slot
                f32x4n* slot = reinterpret_cast<f32x4n*>(&St[wave][lane >> 1][0]);
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    float sv[4], qv[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float a = (float)(__bf16)acc[0][4 * g4 + j], c = (float)(__bf16)acc[1][4 * g4 + j];
                        sv[j] = a + c;
                        qv[j] = fmaf(a, a, c * c);
                        sv[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sv[j]), 0xB1, 0xf, 0xf, true));
                        qv[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, qv[j]), 0xB1, 0xf, 0xf, true));
                    }
                    if (!(lane & 1)) {
                        f32x4n s0 = slot[g4], q0 = slot[4 + g4];
                        s0.x += sv[0]; s0.y += sv[1]; s0.z += sv[2]; s0.w += sv[3];
                        q0.x += qv[0]; q0.y += qv[1]; q0.z += qv[2]; q0.w += qv[3];
                        slot[g4] = s0;
                        slot[4 + g4] = q0;
                    }
                }
            }
        }
        HUPR_STAMP()                                              // 4: tile stored
        if (has_next) {
            if constexpr (!PL) { HUPR_HALO_COMMIT(0) }
            if constexpr (!ABF) {                                 // fp32 sources: second half of the halo, not prefetched
                HUPR_HALO_ISSUE(NH, nxt.b, nxt.tdi * TD, nxt.thi * TH, nxt.twi * TW, nxt.ch * KC)
                HUPR_HALO_COMMIT(NH)
            }
            __syncthreads();
            if constexpr (PL) { HUPR_FRAGS_A(0, 0, 0) }           // its weight fragments were read under the last MFMAs above
        }
        HUPR_STAMP()                                              // 5: next halo in LDS
        cur = nxt;
    }
    if (p.trace != nullptr && tid == 0) p.trace[4096 + 2 * blockIdx.x + 1] = wall_clock64();
    if (p.stats) {
        // channel ch = 32 wn + 4 lh + 8 (r >> 2) + (r & 3) collects, in a fixed order and as doubles, the 16 lane-pair slots of
        // its half-wave lh in each of the four depth-slice waves wm
        __syncthreads();
        if (tid < 2 * BN) {
            const int k = tid >> 6, ch = tid & 63;
            const int wn_ = ch >> 5, c5 = ch & 31, g_ = c5 >> 3, lh_ = (c5 >> 2) & 1, r = 4 * g_ + (c5 & 3);
            double t = 0.0;
#pragma unroll
            for (int wm_ = 0; wm_ < 4; ++wm_)
#pragma unroll
                for (int sl = 0; sl < 16; ++sl) t += (double)St[2 * wm_ + wn_][16 * lh_ + sl][16 * k + r];
            p.stats[(long)blockIdx.x * 2 * p.Co + k * p.Co + ch] = t;
        }
    }
#undef HUPR_FRAGS_A
#undef HUPR_FRAGS_B
#undef HUPR_W_DMA
#undef HUPR_VMCNT
#undef HUPR_STAMP
#undef HUPR_HALO_ISSUE
#undef HUPR_HALO_ISSUE_ITEM
#undef HUPR_HALO_COMMIT
}

static int g_halo_m16 = 1;      // 1 (default): bf16-activation launches go to the 16 x 16 x 32 kernel (conv_halo256m_bf16.hip): -6 % / -7 % on the layer-1 / layer-2 shapes
static int g_halo_m16_td2 = 1;  // its 2 x 8 x 16 tile for D % 4 != 0 (off with hupr_debug_halo_m16(2): 4 x 8 x 8 only)
static int g_halo_m16_2d = 1;   // its 1 x 16 x 16 tile for 1 x 3 x 3 taps (the decoder's convolutions): default since round 5 (built and
                                // parity-tested in round 4, -16...-25 % per launch; hupr_debug_halo_m16(5) = the round-4 default without it)
void set_halo_m16(int on) { g_halo_m16 = on != 0; g_halo_m16_td2 = on != 2; g_halo_m16_2d = (on == 1 || on == 3); }

bool conv_halo256_supported(const HaloArgs& a, int Bn, bool abf) {
    if (a.kd != 3 || a.D % 4 != 0 || a.H % 8 != 0 || a.W % 8 != 0 || a.Ci % 64 != 0 || a.Co % 64 != 0) return false;
    const long tiles = (long)Bn * (a.D / 4) * (a.H / 8) * (a.W / 8) * (a.Co / 64);
    if (tiles >= (1L << 31) || tiles < 256) return false;
    if (abf && (long)Bn * a.D * a.H * a.W * a.in_ld * 2 >= 0x7ffffff0L) return false;
    return true;
}

// Would a launch with fused BatchNorm statistics (a.stats) land on a kernel that has them?  Co = 64: either 256-voxel kernel on the
// 4 x 8 x 8 tile; Co = 128 / 256 (2 / 4 output tiles) or the 2 x 8 x 16 tile: the 16 x 16 x 32 kernel (register-resident sums, round 5).
bool conv_halo256_stats_ok(const HaloArgs& a, int Bn) {
    if (a.kd != 3 || a.Ci % 64 != 0 || a.Co % 64 != 0) return false;
    const int n = a.Co / 64;
    if ((long)Bn * a.D * a.H * a.W * a.in_ld * 2 >= 0x7ffffff0L) return false;
    const bool m16 = g_halo_m16 && a.ablate == 0 && a.trace == nullptr;
    long tiles;
    if (a.D % 4 == 0) {
        if (!conv_halo256_supported(a, Bn, true)) return false;
        if (n == 1) return true;                                   // either kernel, its one-tile form
        if (!m16) return false;
        tiles = (long)Bn * (a.D / 4) * (a.H / 8) * (a.W / 8) * n;
    } else {
        if (!(m16 && g_halo_m16_td2) || a.D % 2 != 0 || a.H % 8 != 0 || a.W % 16 != 0) return false;
        tiles = (long)Bn * (a.D / 2) * (a.H / 8) * (a.W / 16) * n;
        return tiles == 256;                                       // this tile: exactly one tile per workgroup (encoder level 3 at B = 32)
    }
    // a workgroup's run of consecutive tiles must touch at most two distinct output tiles (the register-resident sums)
    const long per_wg = (tiles + kHalo256Grid - 1) / kHalo256Grid;
    return n <= 2 || per_wg == 1;
}

bool launch_conv_halo256(HaloArgs a, int Bn, bool abf, hipStream_t s) {
    // measured (scripts/halo_ablation.py): faster on the 3-D encoder layers, neutral to slower on the 2-D decoder maps
    if (abf && g_halo_m16 && g_halo_m16_td2 && a.kd == 3 && a.D % 4 != 0 && a.D % 2 == 0 && a.H % 8 == 0 && a.W % 16 == 0 && a.Ci % 64 == 0 &&
        a.Co % 64 == 0 && a.ablate == 0 && a.trace == nullptr) {
        // depth not a multiple of four (encoder level 3: D = 2): the 2 x 8 x 16 tile of the 16 x 16 x 32 kernel
        a.TD = 2;
        a.log2TW = 4;
        a.nd = a.D / 2;
        a.nh = a.H / 8;
        a.nw = a.W / 16;
        a.n_co_tiles = a.Co / 64;
        const long tiles2 = (long)Bn * a.nd * a.nh * a.nw * a.n_co_tiles;
        if (tiles2 >= 256 && tiles2 < (1L << 31) && (long)Bn * a.D * a.H * a.W * a.in_ld * 2 < 0x7ffffff0L) {
            launch_conv_halo256m(a, s);
            return true;
        }
    }
    if (abf && g_halo_m16 && g_halo_m16_2d && a.kd == 1 && a.D == 1 && a.H % 16 == 0 && a.W % 16 == 0 && a.Ci % 64 == 0 && a.Co % 64 == 0 &&
        a.ablate == 0 && a.trace == nullptr && !a.stats) {
        // 1 x 3 x 3 convolutions of the decoder: the 1 x 16 x 16 tile of the 16 x 16 x 32 kernel
        a.TD = 1;
        a.log2TW = 4;
        a.nd = 1;
        a.nh = a.H / 16;
        a.nw = a.W / 16;
        a.n_co_tiles = a.Co / 64;
        const long tiles1 = (long)Bn * a.nh * a.nw * a.n_co_tiles;
        if (tiles1 >= 256 && tiles1 < (1L << 31) && (long)Bn * a.H * a.W * a.in_ld * 2 < 0x7ffffff0L) {
            launch_conv_halo256m(a, s);
            return true;
        }
    }
    if (a.kd != 3 || a.D % 4 != 0 || a.H % 8 != 0 || a.W % 8 != 0 || a.Ci % 64 != 0 || a.Co % 64 != 0) return false;
    a.TD = 4;
    a.log2TW = 3;
    a.nd = a.D / 4;
    a.nh = a.H / 8;
    a.nw = a.W / 8;
    a.n_co_tiles = a.Co / 64;
    const long tiles = (long)Bn * a.nd * a.nh * a.nw * a.n_co_tiles;
    if (tiles >= (1L << 31) || tiles < 256) return false;          // small problems: the 128-voxel kernel fills the chip better
    if (abf && (long)Bn * a.D * a.H * a.W * a.in_ld * 2 >= 0x7ffffff0L) return false;   // 32-bit buffer offsets
    // one persistent workgroup per CU
    if (abf && (a.ablate >> 4)) {      // compile-time ablations (profiling only)
        switch (a.ablate >> 4) {
            case 1: HUPR_LAUNCH((hupr_k_conv_halo256_bf16<true, 1>), dim3(kHalo256Grid), dim3(512), 0, s, a); break;
            case 2: HUPR_LAUNCH((hupr_k_conv_halo256_bf16<true, 2>), dim3(kHalo256Grid), dim3(512), 0, s, a); break;
            case 3: HUPR_LAUNCH((hupr_k_conv_halo256_bf16<true, 3>), dim3(kHalo256Grid), dim3(512), 0, s, a); break;
            case 4: HUPR_LAUNCH((hupr_k_conv_halo256_bf16<true, 4>), dim3(kHalo256Grid), dim3(512), 0, s, a); break;
            case 5: HUPR_LAUNCH((hupr_k_conv_halo256_bf16<true, 5>), dim3(kHalo256Grid), dim3(512), 0, s, a); break;
            case 8: HUPR_LAUNCH((hupr_k_conv_halo256_bf16<true, 8>), dim3(kHalo256Grid), dim3(512), 0, s, a); break;
            case 16: HUPR_LAUNCH((hupr_k_conv_halo256_bf16<true, 16>), dim3(kHalo256Grid), dim3(512), 0, s, a); break;
            case 32: HUPR_LAUNCH((hupr_k_conv_halo256_bf16<true, 32>), dim3(kHalo256Grid), dim3(512), 0, s, a); break;
            default: HUPR_LAUNCH((hupr_k_conv_halo256_bf16<true, 7>), dim3(kHalo256Grid), dim3(512), 0, s, a); break;
        }
        return true;
    }
    if (abf && g_halo_m16 && a.ablate == 0 && a.trace == nullptr) launch_conv_halo256m(a, s);      // v_mfma_f32_16x16x32_bf16 form
    else if (abf) HUPR_LAUNCH(hupr_k_conv_halo256_bf16<true>, dim3(kHalo256Grid), dim3(512), 0, s, a);
    else HUPR_LAUNCH(hupr_k_conv_halo256_bf16<false>, dim3(kHalo256Grid), dim3(512), 0, s, a);
    return true;
}

}  // namespace hupr
