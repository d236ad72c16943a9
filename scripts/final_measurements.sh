# usage (GPU box, repo root): bash scripts/final_measurements.sh  -> gpurun_out/{b_*.json, final_kernels.md, gemm_path_calls.txt, tmerge.txt}
python bench.py > gpurun_out/b_default.json 2>gpurun_out/b_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/b_driver.json 2>/dev/null
python bench.py --two-streams --no-cpu-baseline --no-parity-path --no-c2 > gpurun_out/b_two.json 2>/dev/null
python bench.py --workload c2 > gpurun_out/b_c2.json 2>/dev/null
python bench.py --workload c2 --batch 8 > gpurun_out/b_c2b8.json 2>/dev/null
bash scripts/prof_bench.sh final --steps 3 --warmup 2 --no-parity-path --no-cpu-baseline --no-c2 --sustain 0
python scripts/gemm_path_calls.py > gpurun_out/gemm_path_calls.txt 2>&1
python scripts/tmerge_microbench.py 32 > gpurun_out/tmerge.txt 2>&1
HUPR_NO_TMERGE_STREAM=1 python scripts/tmerge_microbench.py 32 >> gpurun_out/tmerge.txt 2>&1
for f in gpurun_out/b_*.json; do tail -1 $f | cut -c1-150; done
