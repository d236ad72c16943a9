#!/bin/bash
# Training step, interleaved on one box: 32x32x16 convolution kernels (HUPR_HALO_M16=0), the 16x16x32 kernel on its 4x8x8 tile only (=2),
# the default (=1: also the 2x8x16 tile for depth-2 layers) and =3 (also the opt-in 1x16x16 tile for the decoder's 1x3x3 taps).  usage: bash scripts/halo_m16_step_ab.sh [rounds] [steps]
rounds=${1:-2}; steps=${2:-60}
for r in $(seq 1 $rounds); do
  for m in 0 2 1 3; do
    v=$(HUPR_HALO_M16=$m python bench.py --steps $steps --warmup 8 --no-c2 --no-parity-path --no-cpu-baseline --sustain 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f frames/s %.3f ms roofline %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))")
    echo "round $r HUPR_HALO_M16=$m: $v"
  done
done
