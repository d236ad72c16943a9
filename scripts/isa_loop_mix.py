"""Static instruction mix of a kernel's hottest loop, from the ISA the build keeps (csrc/Makefile: --save-temps=obj -> build/*.s).

    python scripts/isa_loop_mix.py attention_bf16 attn_fwd_pp64ILi0E attn_bwd_dqILi64EDF16b attn_bwd_dkv512

For every kernel of build/<file>-hip-amdgcn-amd-amdhsa-gfx950.s whose mangled name contains one of the patterns: the loop (label ..
backward branch, >= 8 MFMAs) that is densest in MFMAs, its instruction counts per class and its VALU opcodes.  No GPU needed; the counts
per MFMA agree with the SQ counters of the same kernels (profiles/r04b_attn_sq_pmc.txt: 10.8 VALU instructions per MFMA in
the level-1 attention forward).
"""
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "hupr-a-benchmark-for-human-pose-estimation-using-millimeter-wave-radar_amd"


def cls(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu"
    return "other"


def main():
    path = os.path.join(ROOT, PKG, "build", sys.argv[1] + "-hip-amdgcn-amd-amdhsa-gfx950.s")
    S = open(path).read().split("\n")
    starts = {m.group(1): i for i, l in enumerate(S) for m in [re.match(r"^(_ZN4hupr\w+):\s", l)] if m}
    ends = [i for i, l in enumerate(S) if l.startswith(".Lfunc_end")]
    for name, st in starts.items():
        if not any(p in name for p in sys.argv[2:]):
            continue
        body = S[st:min(e for e in ends if e > st)]
        labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        best = None
        for i, l in enumerate(body):
            m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and labels.get(m.group(1), i) < i:
                a = labels[m.group(1)]
                n = sum("v_mfma" in x for x in body[a:i])
                if n >= 8 and (best is None or n * (best[2] - best[1]) > best[0] * (i - a)):      # densest in MFMAs
                    best = (n, a, i)
        if best is None:
            continue
        ops = collections.Counter(l.split()[0] for l in (x.strip() for x in body[best[1]:best[2]])
                                  if l and not l.startswith((";", ".")))
        per = collections.Counter()
        for op, n in ops.items():
            per[cls(op)] += n
        print("%s\n  loop of %d instructions: %s; VALU per MFMA %.1f" %
              (name, sum(ops.values()), dict(per), per["valu"] / max(per["mfma"], 1)))
        print("  VALU: " + ", ".join("%d %s" % (n, op) for op, n in ops.most_common() if cls(op) == "valu" and n >= 4))


if __name__ == "__main__":
    main()
