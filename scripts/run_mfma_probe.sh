#!/bin/bash
# Build and run scripts/probes/mfma_peak_probe.hip on the GPU box with the shader clock / power sampled next to it
# ("record sclk/power next to mfma_peak_probe").  Output: gpurun_out/<tag>_mfma_peak_probe.txt, and
# profiles/pmc_dominant_kernel.json's measured_ceiling is regenerated FROM that output by scripts/update_ceiling.py.
# usage: bash scripts/run_mfma_probe.sh <tag>
set -e
REPO="$(cd "$(dirname "$0")/.." && pwd)"
TAG="${1:-r03}"
OUT="$REPO/gpurun_out/${TAG}_mfma_peak_probe.txt"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak_probe "$REPO/scripts/probes/mfma_peak_probe.hip"
( for i in $(seq 1 60); do
    echo "--- t=${i}x0.25s"; /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -iE "sclk|power" | head -4; sleep 0.25
  done ) > /tmp/smi_log.txt 2>&1 &
SMI=$!
/tmp/mfma_peak_probe > "$OUT"
kill $SMI 2>/dev/null || true
wait $SMI 2>/dev/null || true
{
  echo
  echo "# rocm-smi samples while the probe ran (every 0.25 s + the tool's own latency): min / max of what was seen"
  grep -i sclk /tmp/smi_log.txt | grep -oE "[0-9]+Mhz" | sort -n | sed -n '1p;$p' | tr '\n' ' ' | sed 's/^/sclk: /'; echo
  grep -i power /tmp/smi_log.txt | grep -oE "[0-9]+\.[0-9]+" | sort -n | sed -n '1p;$p' | tr '\n' ' ' | sed 's/^/power (W): /'; echo
  echo "# raw first / last sample"
  head -5 /tmp/smi_log.txt; tail -4 /tmp/smi_log.txt
} >> "$OUT"
cat "$OUT"
