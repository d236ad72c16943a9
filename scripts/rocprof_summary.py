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
