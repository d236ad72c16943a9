# Round-4 measurement set of the FINAL code (second half of the round; GPU box, repo root): bash scripts/r04b_measurements.sh -> gpurun_out/r04b_*
# bench lines, rocprofv3 kernel summary of the step, PMC passes (FETCH / WRITE; SQ counters) of the dominant convolution and the weight
# gradient on the new stage protocols, halo ablation / trace, GEMM-path calls, PRGCN products, config-5 A/B.
mkdir -p gpurun_out
python bench.py > gpurun_out/r04b_b_default.json 2>gpurun_out/r04b_b_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r04b_b_driver.json 2>/dev/null
python bench.py --two-streams --no-cpu-baseline --no-parity-path --no-c2 > gpurun_out/r04b_b_two.json 2>/dev/null
python bench.py --workload c2 > gpurun_out/r04b_b_c2.json 2>/dev/null
for f in gpurun_out/r04b_b_*.json; do tail -1 $f; done > gpurun_out/r04b_bench_lines.jsonl
bash scripts/prof_bench.sh r04b_bench --steps 3 --warmup 2 --no-parity-path --no-cpu-baseline --no-c2 --sustain 0
bash scripts/pmc_run.sh r04b_conv scripts/pmc_conv1.py
bash scripts/pmc_sq_run.sh r04b_conv scripts/pmc_conv1.py > /dev/null 2>&1
bash scripts/pmc_sq_run.sh r04b_wgrad scripts/pmc_wgrad.py > /dev/null 2>&1
python scripts/halo_ablation.py --bf16act > gpurun_out/r04b_halo_ablation.txt 2>&1
python scripts/halo_trace.py > gpurun_out/r04b_halo_trace.txt 2>&1
python scripts/gemm_path_calls.py > gpurun_out/r04b_gemm_path_calls.txt 2>&1
python scripts/wgrad_microbench.py > gpurun_out/r04b_wgrad_microbench.txt 2>&1
for f in gpurun_out/r04b_b_*.json; do tail -1 $f | cut -c1-160; done
tail -n 3 gpurun_out/r04b_conv_pmc.txt
grep -E "conv_halo256.*(LDS_BANK_CONFLICT|BUSY_CYCLES|WAIT_ANY|WAIT_INST_ANY|ACTIVE_INST_ANY|INSTS_MFMA|WAVE_CYCLES)" gpurun_out/r04b_conv_sq_pmc.txt
