# Round-5 measurement set of the FINAL code (GPU box, repo root): bash scripts/r05_measurements.sh -> gpurun_out/r05_*
# bench lines (default, the driver's form, --attn fp8, --workload c2), rocprofv3 kernel summary of the training steps alone
# (--no-probes: dispatches / 5 = launches per step), SQ counters of the QS attention kernels and the m16 weight gradient,
# attention / weight-gradient A/Bs.
mkdir -p gpurun_out
python bench.py > gpurun_out/r05_b_default.json 2>gpurun_out/r05_b_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_b_driver.json 2>/dev/null
python bench.py --attn fp8 --no-cpu-baseline --no-parity-path --no-c2 > gpurun_out/r05_b_fp8.json 2>/dev/null
python bench.py --workload c2 > gpurun_out/r05_b_c2.json 2>/dev/null
for f in gpurun_out/r05_b_*.json; do tail -1 $f; done > gpurun_out/r05_bench_lines.jsonl
bash scripts/prof_bench.sh r05_bench --steps 3 --warmup 2 --no-parity-path --no-cpu-baseline --no-c2 --sustain 0 --no-probes
bash scripts/prof_bench.sh r05_bench_probes --steps 3 --warmup 2 --no-parity-path --no-cpu-baseline --no-c2 --sustain 0
bash scripts/pmc_sq_run.sh r05_attn scripts/pmc_attn_qs.py > /dev/null 2>&1
bash scripts/pmc_sq_run.sh r05_wgrad scripts/pmc_wgrad.py > /dev/null 2>&1
python scripts/attn_qs_ab.py > gpurun_out/r05_attn_qs_ab.txt 2>&1
python scripts/wgrad_m16_ab.py > gpurun_out/r05_wgrad_m16_ab.txt 2>&1
for f in gpurun_out/r05_b_*.json; do tail -1 $f | cut -c1-160; done
tail -3 gpurun_out/r05_bench_kernels.md
