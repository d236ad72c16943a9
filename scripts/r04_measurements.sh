# Round-4 measurement set (GPU box, repo root): bash scripts/r04_measurements.sh  -> gpurun_out/r04_*
# bench lines, rocprofv3 kernel summary of the step, PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs) of the dominant
# convolution, the FFT chain and the attention forward, the phase ablations / stamps of the attention forward, the halo
# ablation and the MFMA probe with clock / power samples, the FFT cold / warm variants, the GEMM-path calls.
mkdir -p gpurun_out
python bench.py > gpurun_out/r04_b_default.json 2>gpurun_out/r04_b_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_b_driver.json 2>/dev/null
python bench.py --two-streams --no-cpu-baseline --no-parity-path --no-c2 > gpurun_out/r04_b_two.json 2>/dev/null
python bench.py --workload c2 > gpurun_out/r04_b_c2.json 2>/dev/null
python bench.py --workload c2 --batch 8 > gpurun_out/r04_b_c2b8.json 2>/dev/null
for f in gpurun_out/r04_b_*.json; do tail -1 $f; done > gpurun_out/r04_bench_lines.jsonl
bash scripts/prof_bench.sh r04_bench --steps 3 --warmup 2 --no-parity-path --no-cpu-baseline --no-c2 --sustain 0
bash scripts/pmc_run.sh r04_conv scripts/pmc_conv1.py
bash scripts/pmc_run.sh r04_fft scripts/pmc_fft.py
bash scripts/pmc_run.sh r04_attn scripts/pmc_attn.py
bash scripts/pmc_run.sh r04_attn_old scripts/pmc_attn.py 0
python scripts/attn_pp_ablate.py > gpurun_out/r04_attn_ablation.txt 2>&1
python scripts/attn_pp_trace.py > gpurun_out/r04_attn_trace.txt 2>&1
python scripts/halo_ablation.py --bf16act > gpurun_out/r04_halo_ablation.txt 2>&1
bash scripts/run_mfma_probe.sh r04 > /dev/null 2>&1
python scripts/fft_microbench.py 512 cold > gpurun_out/r04_fft_microbench.txt 2>&1
python scripts/gemm_path_calls.py > gpurun_out/r04_gemm_path_calls.txt 2>&1
python scripts/attn_pp_ab.py all > gpurun_out/r04_attn_fwd_ab.txt 2>&1
for f in gpurun_out/r04_b_*.json; do tail -1 $f | cut -c1-160; done
tail -n 3 gpurun_out/r04_conv_pmc.txt gpurun_out/r04_attn_pmc.txt
