"""LDS bank model of the fragment reads of the 16 x 16 x 32 halo convolution (csrc/conv_halo256m_bf16.hip) and of the layout
planned for the weight gradient on the same MFMA form ("What comes next", item 1).  No GPU needed.

Model (MI355X guide, LDS table): 64 banks of 4 bytes; a wave's access is served in fixed lane groups, one LDS cycle per group when
its lanes touch every bank at most once (identical addresses broadcast), one more cycle per further distinct address on a bank.
    ds_read_b128        four groups of sixteen lanes: {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, and the same + 32
    ds_read_b64_tr_b16  two groups of thirty-two lanes
The address formulas below restate the kernels' (file:line in the comments); what the model says is checked against what the
hardware counted: SQ_LDS_BANK_CONFLICT = 0 for the kernel as built (profiles/r04b_conv_sq_pmc.txt) and 32 % of the LDS cycles when
the same lane map ran on the 32 x 32 x 16 kernel's swizzle key — the model gives 0 and 89 % extra cycles on
the activation reads of that combination (about a third of all the kernel's LDS cycles); for the weight gradient as built it gives
0, and 100 % extra on the transpose reads without the one-bit swizzle (measured then: 44 % of the LDS cycles, csrc/wgrad_halo_bf16.hip:46-48).

    python scripts/lds_bank_model.py
"""
import itertools

B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
               [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]
TR64_GROUPS = [list(range(32)), list(range(32, 64))]


def cycles(addr_of_lane, groups, nbytes):
    """(LDS cycles, conflict-free cycles) of one wave instruction: per group, the busiest bank's number of DISTINCT dwords."""
    total = 0
    for g in groups:
        per_bank = {}
        for l in g:
            a = addr_of_lane(l)
            for d in range(nbytes // 4):
                dw = a // 4 + d
                per_bank.setdefault(dw % 64, set()).add(dw)
        total += max(len(s) for s in per_bank.values())
    return total, len(groups)


# ---- csrc/conv_halo256m_bf16.hip: activation fragments (HUPR_XF, :143-146) and weight fragments (woff, :130-137) -------------------
def conv_x_addr(lane, HW, HH, xw0, yw0, dzw, st, rho, kk, key):
    idx, kq = lane & 15, lane >> 4
    yy, wx = idx >> 3, idx & 7
    hz, hy, hx = dzw + st // 3, yw0 + yy + rho, xw0 + wx + st % 3
    chunk = (4 * kk + kq) ^ key(hx, hy)
    return ((hz * HH + hy) * HW + hx) * 128 + chunk * 16


def conv_w_addr(lane, wn, cg, kk):
    idx, kq = lane & 15, lane >> 4
    n = 32 * wn + 16 * cg + idx
    return n * 128 + (((4 * kk + kq) ^ ((n >> 1) & 7)) << 4)


KEY_M16 = lambda hx, hy: ((hx >> 1) & 3) << 1                                  # conv_halo256m_bf16.hip:100
KEY_32 = lambda hx, hy: ((hx >> 1) & 3) | (((hy >> 1) & 1) << 2)               # conv_halo256_bf16.hip:95


def conv_report():
    tiles = {"4 x 8 x 8": (10, 10, [(0, 0)], 3), "2 x 8 x 16": (18, 10, [(0, 0), (8, 0)], 3), "1 x 16 x 16": (18, 18, [(0, 0), (8, 0), (0, 8), (8, 8)], 1)}
    for name, (HW, HH, origins, KD) in tiles.items():
        for label, key in (("kernel's key", KEY_M16), ("32 x 32 x 16 kernel's key", KEY_32)):
            got = want = 0
            for (xw0, yw0), st, rho, kk in itertools.product(origins, range(3 * KD), range(3), range(2)):
                c, f = cycles(lambda l: conv_x_addr(l, HW, HH, xw0, yw0, 0, st, rho, kk, key), B128_GROUPS, 16)
                got, want = got + c, want + f
            print("conv %-11s activation reads, %-25s: %4d LDS cycles for %4d conflict-free (%.0f %% extra)" %
                  (name, label, got, want, 100.0 * (got - want) / want))
    got = want = 0
    for wn, cg, kk in itertools.product(range(2), range(2), range(2)):
        c, f = cycles(lambda l: conv_w_addr(l, wn, cg, kk), B128_GROUPS, 16)
        got, want = got + c, want + f
    print("conv weight reads                                      : %4d LDS cycles for %4d conflict-free" % (got, want))
    return got == want


# ---- the weight gradient on v_mfma_f32_16x16x32_bf16 (planned): transpose reads of row-major [voxel][64 channels] images -----------
# Lane (s = lane & 15, kq = lane >> 4) of a 16 x 16 x 32 operand owns column s and the reduction elements 8 kq .. 8 kq + 7: two
# ds_read_b64_tr_b16 (t = 0, 1) of four rows each.  Inside a 16-lane group supplier lane s addresses the 8-byte segment 4 (s & 3) of
# row rho[s >> 2] (csrc/wgrad_halo_bf16.hip:6-11).  A 32-lane group = two values of kq = eight rows x one 32-byte column segment.
def wgrad_addr(lane, t, rows_of, col_seg32, row_shift, swz):
    s, kq = lane & 15, lane >> 4
    row = rows_of(kq, t, s >> 2) + row_shift
    col = 32 * col_seg32 + 8 * (s & 3)                     # byte column inside the 128-byte row
    return row * 128 + swz(row, col)


SWZ_NOW = lambda row, col: col ^ (((row >> 1) & 1) << 6)                       # wgrad_halo_bf16.hip:49 (one bit, 64-byte halves)
SWZ_PLAN = lambda row, col: col ^ (((row >> 1) & 3) << 5)                      # two bits on 32-byte granules


def wgrad_now_addr(lane, t, kx, base_row, wn, swz, HW=10):
    """The kernel as built (32 x 32 x 16 operands; wgrad_halo_bf16.hip:73-92): 16-lane group g = lane >> 4 serves column half g & 1
    of reduction half g >> 1; a 32-lane group reads four rows x 64 bytes."""
    g, s = lane >> 4, lane & 15
    c = 8 * (g >> 1) + 4 * t + (s >> 2)
    row = (c >> 3) * HW + (c & 7) + kx + base_row
    return row * 128 + swz(row, wn * 64 + (16 * (g & 1) + 4 * (s & 3)) * 2)


def wgrad_report():
    HW = 10
    for sname, swz in (("one-bit swizzle (as built)", SWZ_NOW), ("no swizzle (round 2)", lambda row, col: col)):
        got = want = 0
        for base, kx, wn, t in itertools.product(range(0, 4 * HW, 2), range(3), range(2), range(2)):
            c, f = cycles(lambda l: wgrad_now_addr(l, t, kx, base, wn, swz), TR64_GROUPS, 8)
            got, want = got + c, want + f
        print("wgrad 32 x 32 x 16 x halo reads, %-27s: %5d LDS cycles for %5d conflict-free (%.0f %% extra)" %
              (sname, got, want, 100.0 * (got - want) / want))
    # voxel owned by (kq, t, j): tile column 4 (kq & 1) + j, tile row 2 (kq >> 1) + t  ->  image row = row * HW + column (x halo) / 8 * row + column (dy)
    own = {"x halo": lambda kq, t, j: (2 * (kq >> 1) + t) * HW + 4 * (kq & 1) + j,
           "dy tile": lambda kq, t, j: (2 * (kq >> 1) + t) * 8 + 4 * (kq & 1) + j}
    # the obvious ownership (reduction element 8 kq + 4 t + j = tile voxel in row-major order) for comparison
    naive = {"x halo": lambda kq, t, j: ((8 * kq + 4 * t + j) >> 3) * HW + ((8 * kq + 4 * t + j) & 7),
             "dy tile": lambda kq, t, j: 8 * kq + 4 * t + j}
    ok = True
    for what in ("x halo", "dy tile"):
        for oname, rows in (("planned ownership", own[what]), ("row-major ownership", naive[what])):
            for sname, swz in (("two-bit swizzle", SWZ_PLAN), ("today's one-bit swizzle", SWZ_NOW)):
                got = want = 0
                shifts = range(0, 3 * HW + 3) if what == "x halo" else range(0, 128, 32)      # every tap offset / K-step base
                for shift, seg, t in itertools.product(shifts, range(4), range(2)):
                    c, f = cycles(lambda l: wgrad_addr(l, t, rows, seg, shift, swz), TR64_GROUPS, 8)
                    got, want = got + c, want + f
                print("wgrad m16 %-8s %-20s %-24s: %5d LDS cycles for %5d conflict-free (%.0f %% extra)" %
                      (what, oname, sname, got, want, 100.0 * (got - want) / want))
                if oname == "planned ownership" and sname == "two-bit swizzle":
                    ok = ok and got == want
    return ok


# ---- csrc/attention_bf16.hip: row-major [row][D] bf16 images, 16-byte chunks at position chunk ^ key(row) (Img<D>) -----------------
# b128 fragments (mma_rows_x_frags / HUPR_PP_READ_K: lane reads row 32 t + (lane & 31), chunk 2 ks + (lane >> 5)) and transpose
# fragments (mma_tr_x_tile / HUPR_PP_READ_V: 16-lane group g serves column half g & 1 of row quartet g >> 1).  Rounds 1-4 keyed the
# swizzle with the row bits in place; round 5 rotates them (the comment at Img<D>::key).  Counters for the old key:
# SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 33 % (forward), 20 % (dQ), 22 % (dK / dV) in profiles/r04b_attn_sq_pmc.txt.
ATTN_KEYS = {
    64: (lambda r: (r >> 1) & 7, lambda r: (((r >> 1) & 1) << 2) | ((r >> 2) & 3)),
    128: (lambda r: r & 15, lambda r: ((r & 3) << 2) | ((r >> 2) & 3)),
    256: (lambda r: r & 31, lambda r: ((r & 3) << 2) | ((r >> 2) & 3) | (r & 16)),
}


def attn_cycles(D, key):
    rb = 2 * D
    b = [0, 0]
    for t, ks in itertools.product(range(2), range(D // 16)):
        c, f = cycles(lambda l: (32 * t + (l & 31)) * rb + (((2 * ks + (l >> 5)) ^ key(32 * t + (l & 31))) << 4), B128_GROUPS, 16)
        b[0], b[1] = b[0] + c, b[1] + f
    tr = [0, 0]
    for t, u, ct, second in itertools.product(range(2), range(2), range(D // 32), range(2)):
        def addr(l):
            g, s = l >> 4, l & 15
            row = 32 * t + 16 * u + 4 * (g >> 1) + (s >> 2) + 8 * second
            c = 32 * ct + 16 * (g & 1) + 4 * (s & 3)
            return row * rb + (((c >> 3) ^ key(row)) << 4) + (c & 4) * 2
        c, f = cycles(addr, TR64_GROUPS, 8)
        tr[0], tr[1] = tr[0] + c, tr[1] + f
    return b, tr


def attn_report():
    ok = True
    for D, (old, new) in ATTN_KEYS.items():
        for name, key in (("rounds 1-4 key", old), ("rotated key", new)):
            b, tr = attn_cycles(D, key)
            # per 64-row tile and wave: forward = one image b128 + one transposed; dQ = two b128 + one transposed; dK / dV = two + two
            mix = {"fwd": (1, 1), "dQ": (2, 1), "dK/dV": (2, 2)}
            shares = ", ".join("%s %.0f %%" % (k, 100.0 * (nb * (b[0] - b[1]) + nt * (tr[0] - tr[1])) / (nb * b[0] + nt * tr[0]))
                               for k, (nb, nt) in mix.items())
            print("attention D = %3d %-15s: b128 reads %3d LDS cycles for %3d conflict-free, transpose reads %3d for %3d; conflict share of "
                  "the fragment-read cycles: %s" % (D, name, b[0], b[1], tr[0], tr[1], shares))
            if name == "rotated key":
                ok = ok and b[0] == b[1] and tr[0] == tr[1]
    return ok


if __name__ == "__main__":
    a = conv_report()
    b = wgrad_report()
    c = attn_report()
    print("attention images conflict-free with the rotated key: %s" % c)
    print("conv weights conflict-free: %s; planned weight-gradient layout conflict-free for every tap: %s" % (a, b))
