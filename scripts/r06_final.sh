# Round 6, final tree: full GPU suite (with the printed gate numbers), then the evidence set (scripts/r06_evidence.sh)
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s --durations=12 > gpurun_out/r06_suite_final_full.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" gpurun_out/r06_suite_final_full.txt | tail -20 > gpurun_out/r06_suite_final.txt
grep -E "held-out scenes|vs oracle|path vs|OKS AP|fit:|pose-scene fit|further fit|passed|failed" gpurun_out/r06_suite_final_full.txt | cut -c1-400 > gpurun_out/r06_trained_gates_final.txt
tail -3 gpurun_out/r06_suite_final.txt
bash scripts/r06_evidence.sh
