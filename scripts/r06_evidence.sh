# Round 6 evidence on the round-6 tree (GPU box, repo root): the driver's bench line, a rocprofv3 kernel table of the SAME invocation
# style (--steps 20 --warmup 5), FETCH/WRITE and SQ counter passes of the dominant kernel, the default-length line and the C2 line.
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_b_driver.json 2>gpurun_out/r06_b_driver.err
bash scripts/prof_bench.sh r06_driver --steps 20 --warmup 5
bash scripts/prof_bench.sh r06_bench --steps 3 --warmup 2 --no-parity-path --no-cpu-baseline --no-c2 --sustain 0 --no-probes
bash scripts/pmc_run.sh r06_conv scripts/pmc_conv1.py > /dev/null
bash scripts/pmc_sq_run.sh r06_conv scripts/pmc_conv1.py > /dev/null
python bench.py > gpurun_out/r06_b_default.json 2>/dev/null
python bench.py --workload c2 > gpurun_out/r06_b_c2.json 2>/dev/null
for f in gpurun_out/r06_b_driver.json gpurun_out/r06_b_default.json gpurun_out/r06_b_c2.json; do tail -1 $f; done > gpurun_out/r06_bench_lines.jsonl
tail -1 gpurun_out/r06_b_driver.json | cut -c1-300
cat gpurun_out/r06_conv_pmc.txt
grep -E "conv_halo256m" gpurun_out/r06_driver_kernels.md | head -8 | cut -c1-200
