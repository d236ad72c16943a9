#!/bin/bash
# rocprofv3 --kernel-trace of one bench.py command -> gpurun_out/<tag>_kernels.md (per-kernel table + stream-occupancy line).
# usage (on the GPU box, from the repo root): bash scripts/prof_bench.sh <tag> [bench.py arguments ...]
set -e
REPO="$(cd "$(dirname "$0")/.." && pwd)"
TAG="$1"; shift
export TMPDIR=/tmp
OUT=/tmp/prof_$TAG
rm -rf "$OUT"
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT" -o run -- python "$REPO/bench.py" "$@" > "$REPO/gpurun_out/${TAG}_line.json" 2> /tmp/prof_$TAG.err) || { tail -20 /tmp/prof_$TAG.err; exit 1; }
DB=$(find "$OUT" -name "*.db" | head -1)
python "$REPO/scripts/rocprof_summary.py" "$DB" "$TAG: bench.py $*" > "$REPO/gpurun_out/${TAG}_kernels.md"
python "$REPO/scripts/conv_instep_durations.py" "$DB" > "$REPO/gpurun_out/${TAG}_conv_instep.txt" 2>&1 || true
tail -2 "$REPO/gpurun_out/${TAG}_kernels.md"
