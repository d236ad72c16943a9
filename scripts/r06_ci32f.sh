# round 6: the first layer's forward (Ci = 32) on the 256-voxel kernel's 64-byte-row form + 256 partial tensors for the register-staged 2-D
# weight gradient: parity tests, micro A/B, same-box step A/B against the previous build (HUPR_LIB_PATH), kernel table
mkdir -p gpurun_out
P=hupr-a-benchmark-for-human-pose-estimation-using-millimeter-wave-radar_amd
python -m pytest tests/test_ops_gpu.py tests/test_abi.py -m gpu -q -x -k "wgrad or conv or abi" > gpurun_out/r06_ci32f_tests.txt 2>&1; tail -3 gpurun_out/r06_ci32f_tests.txt
python scripts/conv_ci32_ab.py > gpurun_out/r06_ci32f_micro.txt 2>&1; grep kernel gpurun_out/r06_ci32f_micro.txt
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-30s %8.1f frames/s  %6.3f ms  %4d launches' % (sys.argv[1], d['value'], d['ms_per_step'], d['launches_per_step']))" "$1"; }
A="--steps 60 --warmup 10 --no-cpu-baseline --no-parity-path --no-c2 --sustain 0 --no-probes"
{
for i in 1 2 3; do
HUPR_LIB_PATH=$PWD/$P/lib/libhupr_prev.so python bench.py $A 2>/dev/null | pr "prev"
python bench.py $A 2>/dev/null | pr "new"
done
} > gpurun_out/r06_ci32f_ab.txt
cat gpurun_out/r06_ci32f_ab.txt
bash scripts/prof_bench.sh r06g_bench --steps 3 --warmup 2 --no-parity-path --no-cpu-baseline --no-c2 --sustain 0 --no-probes > /dev/null 2>&1
grep -E "conv_halo|wgrad_halo_bf16|splitk" gpurun_out/r06g_bench_kernels.md | cut -c1-200 | head -40
