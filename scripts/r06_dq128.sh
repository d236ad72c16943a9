# round 6: level-2 dQ kernel on 32-key tiles (<= 256 registers) + deferred-bias epilogue of the 32-input-channel convolution: tests, checksums
# of both builds (identical arithmetic claimed), conv micro A/B, same-box step A/B (prev = the same tree with the previous attention kernels)
mkdir -p gpurun_out
P=hupr-a-benchmark-for-human-pose-estimation-using-millimeter-wave-radar_amd
python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "attn or attention or mscsa or first_layer" > gpurun_out/r06_dq64_tests.txt 2>&1; tail -2 gpurun_out/r06_dq64_tests.txt
{ echo "# new"; python scripts/attn_kernels_time.py 2>/dev/null | grep "N="; echo "# prev"; HUPR_LIB_PATH=$PWD/$P/lib/libhupr_prev.so python scripts/attn_kernels_time.py 2>/dev/null | grep "N="; } > gpurun_out/r06_dq64_kernels.txt; cat gpurun_out/r06_dq64_kernels.txt
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-30s %8.1f frames/s  %6.3f ms  %4d launches' % (sys.argv[1], d['value'], d['ms_per_step'], d['launches_per_step']))" "$1"; }
A="--steps 60 --warmup 10 --no-cpu-baseline --no-parity-path --no-c2 --sustain 0 --no-probes"
{
for i in 1 2 3; do
HUPR_LIB_PATH=$PWD/$P/lib/libhupr_prev.so python bench.py $A 2>/dev/null | pr "prev (dQ<64>: 64-key tiles, 208 registers)"
python bench.py $A 2>/dev/null | pr "new (32-key tiles, 156 registers)"
done
} > gpurun_out/r06_dq64_ab.txt
cat gpurun_out/r06_dq64_ab.txt
bash scripts/prof_bench.sh r06j_bench --steps 3 --warmup 2 --no-parity-path --no-cpu-baseline --no-c2 --sustain 0 --no-probes > /dev/null 2>&1
grep -E "attn|conv_halo256m_bf16<4, 8, 8, 3, 0, 2, 32>" gpurun_out/r06j_bench_kernels.md | cut -c1-200
