# round 6, step 4: same-box A/B of library builds (HUPR_LIB_PATH): prev = the pruned round-5 kernels; new = split-K row reduce (4-slice class), 6 / 3 rows in flight in the
# BatchNorm backward statistics, 16 partial rows in flight in the finalizes
mkdir -p gpurun_out
P=hupr-a-benchmark-for-human-pose-estimation-using-millimeter-wave-radar_amd
python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "splitk or batchnorm or bn_ or norm" > gpurun_out/r06_step4_tests.txt 2>&1; tail -3 gpurun_out/r06_step4_tests.txt
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-30s %8.1f frames/s  %6.3f ms  %4d launches' % (sys.argv[1], d['value'], d['ms_per_step'], d['launches_per_step']))" "$1"; }
A="--steps 60 --warmup 10 --no-cpu-baseline --no-parity-path --no-c2 --sustain 0 --no-probes"
{
for i in 1 2 3; do
HUPR_LIB_PATH=$P/lib/libhupr_prev.so python bench.py $A 2>/dev/null | pr "prev (pruned r05 kernels)"
python bench.py $A 2>/dev/null | pr "new"
done
} > gpurun_out/r06_step4_ab.txt
cat gpurun_out/r06_step4_ab.txt
bash scripts/prof_bench.sh r06d_bench --steps 3 --warmup 2 --no-parity-path --no-cpu-baseline --no-c2 --sustain 0 --no-probes
HUPR_LIB_PATH=$P/lib/libhupr_prev.so bash scripts/prof_bench.sh r06d_prev --steps 3 --warmup 2 --no-parity-path --no-cpu-baseline --no-c2 --sustain 0 --no-probes
for t in r06d_prev r06d_bench; do echo $t; for k in splitk_reduce colstats finalize; do grep -E "$k" gpurun_out/${t}_kernels.md | awk -F'|' -v k=$k '{s+=$3; c+=$4} END {print "  " k, s/10, "ms/step", c/10, "launches/step"}'; done; done
