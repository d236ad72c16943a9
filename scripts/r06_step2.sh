# round 6, step 2: GPU suite on the pruned tree + split-K reduction A/B (same box, interleaved)
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --durations=15 -x > gpurun_out/r06_suite_step2.txt 2>&1
tail -5 gpurun_out/r06_suite_step2.txt
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-30s %8.1f frames/s  %6.3f ms  %4d launches' % (sys.argv[1], d['value'], d['ms_per_step'], d['launches_per_step']))" "$1"; }
A="--steps 60 --warmup 10 --no-cpu-baseline --no-parity-path --no-c2 --sustain 0 --no-probes"
cat > /tmp/legacy_reduce.py <<'PY'
import runpy, sys
sys.path.insert(0, '.')
from hupr_amd import runtime
runtime.lib().hupr_debug_splitk_slices(256)
sys.argv = ['bench.py'] + sys.argv[1:]
runpy.run_path('bench.py', run_name='__main__')
PY
{
for i in 1 2 3; do
python /tmp/legacy_reduce.py $A 2>/dev/null | pr "scattered-store reduce"
python bench.py $A 2>/dev/null | pr "LDS-transposed reduce"
done
} > gpurun_out/r06_splitk_ab.txt
cat gpurun_out/r06_splitk_ab.txt
bash scripts/prof_bench.sh r06b_bench --steps 3 --warmup 2 --no-parity-path --no-cpu-baseline --no-c2 --sustain 0 --no-probes
grep -E "splitk_reduce" gpurun_out/r06b_bench_kernels.md | awk -F'|' '{s+=$3; c+=$4} END {print s/10, "ms/step", c/10, "launches/step"}'
