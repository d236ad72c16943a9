python bench.py > gpurun_out/r04_b_default.json 2>gpurun_out/r04_b_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_b_driver.json 2>/dev/null
python bench.py --two-streams --no-cpu-baseline --no-parity-path --no-c2 > gpurun_out/r04_b_two.json 2>/dev/null
for f in gpurun_out/r04_b_default.json gpurun_out/r04_b_driver.json gpurun_out/r04_b_two.json gpurun_out/r04_b_c2.json gpurun_out/r04_b_c2b8.json; do tail -n 1 $f; done > gpurun_out/r04_bench_lines.jsonl
bash scripts/prof_bench.sh r04_bench --steps 3 --warmup 2 --no-parity-path --no-cpu-baseline --no-c2 --sustain 0
bash scripts/pmc_run.sh r04_attn_old scripts/pmc_attn.py 0 > /dev/null
python scripts/attn_pp_ab.py all > gpurun_out/r04_attn_fwd_ab.txt 2>&1
tail -n 6 gpurun_out/r04_attn_fwd_ab.txt
cat gpurun_out/r04_attn_old_pmc.txt
for f in gpurun_out/r04_b_default.json gpurun_out/r04_b_driver.json gpurun_out/r04_b_two.json; do tail -n 1 $f | cut -c1-200; done
tail -n 3 gpurun_out/r04_b_default.err
