"""Per-launch durations of the dominant convolution kernel inside the traced training steps, each with the kernel that ran right before
it and the idle gap in front: does the in-step average (215-226 us) exceed the microbenchmark (188-204 us) on particular launches?
usage: python scripts/conv_instep_durations.py <results.db>"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "").replace("hupr::", "")[:60]
big = [i for i, r in enumerate(rows) if "conv_halo256" in r[0] and r[2] - r[1] > 150e3]
print("%d layer-1-shaped launches of hupr_k_conv_halo256_bf16" % len(big))
by_prev = {}
for i in big:
    prev = short(rows[i - 1][0]) if i else "-"
    by_prev.setdefault(prev, []).append(((rows[i][2] - rows[i][1]) / 1e3, (rows[i][1] - rows[i - 1][2]) / 1e3 if i else 0.0))
for prev, v in sorted(by_prev.items(), key=lambda kv: -len(kv[1])):
    d = [x[0] for x in v]
    print("  after %-62s %3d launches: %.1f .. %.1f us, mean %.1f (gap in front %.1f us)" % (prev, len(d), min(d), max(d), sum(d) / len(d), sum(x[1] for x in v) / len(v)))
