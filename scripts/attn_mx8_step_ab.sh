#!/bin/bash
# Config 5 as a TRAINING configuration, measured: the step with the block-scaled fp8 forward of the level-1 MSCSA attentions
# (HUPR_ATTN_FP8=mx; backward on the bf16 kernels) against the default step, interleaved on one box.  usage: bash scripts/attn_mx8_step_ab.sh [rounds] [steps]
rounds=${1:-2}; steps=${2:-60}
for r in $(seq 1 $rounds); do
  a=$(python bench.py --steps $steps --warmup 8 --no-c2 --no-parity-path --no-cpu-baseline --sustain 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f frames/s %.3f ms loss %.5f' % (d['value'], d['ms_per_step'], d['loss']))")
  b=$(HUPR_ATTN_FP8=mx python bench.py --steps $steps --warmup 8 --no-c2 --no-parity-path --no-cpu-baseline --sustain 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f frames/s %.3f ms loss %.5f' % (d['value'], d['ms_per_step'], d['loss']))")
  echo "round $r: bf16 attention $a | HUPR_ATTN_FP8=mx $b"
done
