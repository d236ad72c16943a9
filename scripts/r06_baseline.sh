# Round 6 baseline on the unchanged round-5 tree: driver-form bench line + rocprofv3 kernel table of the training steps alone.
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/r06a_b_driver.json 2>gpurun_out/r06a_b_driver.err
bash scripts/prof_bench.sh r06a_bench --steps 3 --warmup 2 --no-parity-path --no-cpu-baseline --no-c2 --sustain 0 --no-probes
tail -1 gpurun_out/r06a_b_driver.json | cut -c1-400
