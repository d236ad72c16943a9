"""Summarise a rocprofv3 --kernel-trace results.db into a per-kernel table (markdown).
usage: python scripts/rocprof_summary.py <results.db> [title] > profiles/<name>.md"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                  "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("# %s\n" % title)
print("total kernel time %.3f ms over %d dispatches\n" % (tot / 1e6, sum(r[1] for r in rows)))
print("| % | total ms | calls | avg us | min us | max us | vgpr | agpr | lds B | kernel |")
print("|---|---|---|---|---|---|---|---|---|---|")
for r in rows:
    name = re.sub(r"\(.*", "", r[0]).replace("void ", "")[:120]
    print("| %.2f | %.3f | %d | %.1f | %.1f | %.1f | %s | %s | %s | `%s` |" % (100 * r[2] / tot, r[2] / 1e6, r[1], r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, r[6], r[7], r[8], name))

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
