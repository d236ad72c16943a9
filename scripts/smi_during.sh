#!/bin/bash
# Shader clock / socket power sampled (rocm-smi, every ~0.25 s) while a command runs: is a kernel loop at the chip's power limit?
# usage (GPU box, repo root): bash scripts/smi_during.sh <tag> <command ...>   -> gpurun_out/<tag>_smi.txt
REPO="$(cd "$(dirname "$0")/.." && pwd)"
TAG="$1"; shift
OUT="$REPO/gpurun_out/${TAG}_smi.txt"
( while true; do
    /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -iE "sclk|Socket Graphics" | head -2 | tr '\n' ' '; echo; sleep 0.2
  done ) > /tmp/smi_$TAG.txt 2>&1 &
SMI=$!
"$@" > /tmp/smi_cmd_$TAG.txt 2>&1
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
{
  echo "# $*"
  tail -3 /tmp/smi_cmd_$TAG.txt
  echo "# samples: sclk MHz / power W (all, in time order)"
  grep -oE "\([0-9]+Mhz\)|[0-9]+\.[0-9]+$|\(W\): [0-9.]+" /tmp/smi_$TAG.txt | tr -d '()' | sed 's/W: //' | paste - - | tr '\n' ';'
  echo
} > "$OUT"
cat "$OUT"
