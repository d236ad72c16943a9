#!/bin/bash
# Interleaved same-box A/B of two builds of the library on the training step: scripts/lib_ab.sh <other .so> [rounds] [steps]
# prints frames/s of `python bench.py --steps N` for the in-tree library and for HUPR_LIB_PATH=<other>, alternating.
other=$1; rounds=${2:-3}; steps=${3:-60}
for r in $(seq 1 $rounds); do
  a=$(python bench.py --steps $steps --warmup 8 --no-c2 --no-parity-path --no-cpu-baseline --sustain 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f frames/s %.3f ms' % (d['value'], d['ms_per_step']))")
  b=$(HUPR_LIB_PATH=$other python bench.py --steps $steps --warmup 8 --no-c2 --no-parity-path --no-cpu-baseline --sustain 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f frames/s %.3f ms' % (d['value'], d['ms_per_step']))")
  echo "round $r: in-tree $a | $other $b"
done
