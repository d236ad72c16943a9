"""Regenerate `measured_ceiling_tflops` of profiles/pmc_dominant_kernel.json from a committed mfma_peak_probe output, so that
bench.py's `frac_of_measured_ceiling` is reproducible from profiles/ (the JSON said 1 475, the probe file 1 503).
usage: python scripts/update_ceiling.py profiles/r03_mfma_peak_probe.txt"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
txt = open(src).read()
m = re.search(r"MFMA \+ 7 ds_read_b128 per 6 MFMAs.*?(\d+) TF/s", txt)
r = re.search(r"random operands\s+NACC 2, 8 waves/CU.*?(\d+) TF/s", txt)
c = re.search(r"constant operands NACC 8, 4 waves/CU.*?(\d+) TF/s", txt)
assert m, "probe output has no '7 ds_read_b128 per 6 MFMAs' line"
p = os.path.join(ROOT, "profiles", "pmc_dominant_kernel.json")
d = json.load(open(p))
rel = os.path.relpath(os.path.abspath(src), ROOT)
d["bf16"]["measured_ceiling_tflops"] = float(m.group(1))
d["bf16"]["measured_ceiling_note"] = ("v_mfma_f32_32x32x16_bf16 with 7 ds_read_b128 per 6 MFMAs (this kernel's ratio), random operands from "
                                      "LDS, no barriers, no global traffic, all 256 CUs: %s TF/s; register-only random operands %s, "
                                      "constant operands %s (scripts/probes/mfma_peak_probe.hip, %s — with the sclk / power samples "
                                      "taken while it ran)" % (m.group(1), r.group(1) if r else "?", c.group(1) if c else "?", rel))
d["bf16"]["limiter"] = re.sub(r"\d[ \d]*TF/s with this kernel's 7 ds_read_b128 per 6 MFMAs", "%s TF/s with this kernel's 7 ds_read_b128 per 6 MFMAs" % m.group(1),
                              d["bf16"]["limiter"]).replace("profiles/r02_mfma_peak_probe.txt", rel)
json.dump(d, open(p, "w"), indent=1)
print("measured_ceiling_tflops =", m.group(1), "from", rel)
