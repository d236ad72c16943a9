"""Summarise a rocprofv3 --kernel-trace results.db into a per-kernel table (markdown).
usage: python scripts/rocprof_summary.py <results.db> [title] > profiles/<name>.md"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
# One row per (kernel, launch geometry, duration cluster).  A kernel name that serves several layer shapes is split by grid /
# workgroup / LDS size; persistent kernels launch one workgroup per CU for EVERY shape, so their launches are further split
# where the sorted durations jump by more than 1.3x (the layer shapes are >= 2x apart in work): the cluster of the largest
# shape is then directly comparable with bench.py's per-launch average for that shape.
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
geo = [c for c in ("grid_size", "grid_size_x", "grid_x", "workgroup_size", "workgroup_size_x", "workgroup_x") if c in cols]
sel = "select name, end-start, vgpr_count, accum_vgpr_count, lds_size" + "".join(", " + c for c in geo) + " from kernels"
groups = {}
for r in db.execute(sel):
    groups.setdefault((r[0], r[4]) + tuple(r[5:]), []).append(r)
rows = []
for key, rs in groups.items():
    durs = sorted(x[1] for x in rs)
    cuts = [0] + [i for i in range(1, len(durs)) if durs[i] > 1.3 * durs[i - 1] and durs[i] > 20000] + [len(durs)]
    persistent = len(cuts) > 2 and len(rs) >= 8
    parts = [durs[a:b] for a, b in zip(cuts[:-1], cuts[1:])] if persistent else [durs]
    for k, d in enumerate(parts):
        tag = " [duration cluster %d/%d]" % (k + 1, len(parts)) if len(parts) > 1 else ""
        rows.append((key[0] + tag, len(d), sum(d), sum(d) / len(d), d[0], d[-1], max(x[2] for x in rs), max(x[3] for x in rs), key[1],
                     "/".join(str(v) for v in key[2:])))
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
print("# %s\n" % title)
print("total kernel time %.3f ms over %d dispatches\n" % (tot / 1e6, sum(r[1] for r in rows)))
print("| % | total ms | calls | avg us | min us | max us | vgpr | agpr | lds B | grid / wg | kernel |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for r in rows:
    name = re.sub(r"\((?!duration).*?(?= \[duration|$)", "", r[0]).replace("void ", "")[:150]
    print("| %.2f | %.3f | %d | %.1f | %.1f | %.1f | %s | %s | %s | %s | `%s` |" % (100 * r[2] / tot, r[2] / 1e6, r[1], r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, r[6], r[7], r[8], r[9], name))

# ---- stream occupancy: how much of the traced span had no kernel running (launch gaps / host-bound stretches) ----
iv = db.execute("select start, end from kernels order by start").fetchall()
if iv:
    busy, cur_s, cur_e, gaps = 0, iv[0][0], iv[0][1], []
    for s_, e_ in iv[1:]:
        if s_ > cur_e:
            busy += cur_e - cur_s
            gaps.append(s_ - cur_e)
            cur_s, cur_e = s_, e_
        else:
            cur_e = max(cur_e, e_)
    busy += cur_e - cur_s
    span = iv[-1][1] - iv[0][0]
    big = [g for g in gaps if g > 1e6]            # > 1 ms: step boundaries / warm-up, excluded from the "inner" idle figure
    inner = [g for g in gaps if g <= 1e6]
    print("\nstream occupancy: span %.1f ms, kernels running %.1f ms; idle inside steps %.1f ms in %d gaps (median %.1f us, "
          "%d gaps > 10 us totalling %.1f ms); %d long gaps (> 1 ms) totalling %.1f ms" %
          (span / 1e6, busy / 1e6, sum(inner) / 1e6, len(inner), (sorted(inner)[len(inner) // 2] / 1e3 if inner else 0),
           sum(1 for g in inner if g > 1e4), sum(g for g in inner if g > 1e4) / 1e6, len(big), sum(big) / 1e6))
