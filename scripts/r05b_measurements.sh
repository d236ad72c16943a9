# Round 5, second half: measurement set of the FINAL code (GPU box, repo root): bash scripts/r05b_measurements.sh -> gpurun_out/r05b_*
# bench lines (default, the driver's form, --workload c2), rocprofv3 kernel summary of the training steps alone (--no-probes: dispatches / 5
# = launches per step), and the second half's launch merges switched off / on in interleaved runs of the same box.
mkdir -p gpurun_out
python bench.py > gpurun_out/r05b_b_default.json 2>gpurun_out/r05b_b_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r05b_b_driver.json 2>/dev/null
python bench.py --workload c2 > gpurun_out/r05b_b_c2.json 2>/dev/null
for f in gpurun_out/r05b_b_*.json; do tail -1 $f; done > gpurun_out/r05b_bench_lines.jsonl
bash scripts/prof_bench.sh r05b_bench --steps 3 --warmup 2 --no-parity-path --no-cpu-baseline --no-c2 --sustain 0 --no-probes
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-44s %8.1f frames/s  %6.3f ms  %4d launches' % (sys.argv[1], d['value'], d['ms_per_step'], d['launches_per_step']))" "$1"; }
{
echo "python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-parity-path --no-c2 --sustain 0 --no-probes, interleaved on one box"
for i in 1 2 3; do
HUPR_NO_ATTN_LEVEL_BATCH=1 HUPR_NO_BN_FINALIZE_PAIR=1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-parity-path --no-c2 --sustain 0 --no-probes 2>/dev/null | pr "per-attention launches, single finalizes"
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-parity-path --no-c2 --sustain 0 --no-probes 2>/dev/null | pr "level-wide launches, paired finalize (default)"
done
} > gpurun_out/r05b_launch_merges_ab.txt
{
echo "python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-parity-path --no-c2 --sustain 0 --no-probes, interleaved on one box"
echo "off = HUPR_NO_ATTN_LEVEL_BATCH HUPR_NO_BN_FINALIZE_PAIR HUPR_NO_DUAL_WGRAD HUPR_NO_PAIR_BCE HUPR_NO_RES_PREFETCH = 1 (the kernels of the round's first half)"
for i in 1 2 3; do
HUPR_NO_ATTN_LEVEL_BATCH=1 HUPR_NO_BN_FINALIZE_PAIR=1 HUPR_NO_DUAL_WGRAD=1 HUPR_NO_PAIR_BCE=1 HUPR_NO_RES_PREFETCH=1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-parity-path --no-c2 --sustain 0 --no-probes 2>/dev/null | pr "second half of round 5 off"
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-parity-path --no-c2 --sustain 0 --no-probes 2>/dev/null | pr "default"
done
} > gpurun_out/r05b_second_half_ab.txt
for f in gpurun_out/r05b_b_*.json; do tail -1 $f | cut -c1-160; done
cat gpurun_out/r05b_launch_merges_ab.txt gpurun_out/r05b_second_half_ab.txt
tail -3 gpurun_out/r05b_bench_kernels.md
