#!/bin/bash
# HBM bytes per launch from rocprofv3 PMC counters, collected exactly as /opt/skills/guides/MI355X_MICROARCH.md prescribes:
# one pass per counter, --pmc together with --kernel-trace only.  -> gpurun_out/<tag>_pmc.txt
# usage (GPU box, repo root): bash scripts/pmc_run.sh <tag> <python script> [args...]
set -e
REPO="$(cd "$(dirname "$0")/.." && pwd)"
TAG="$1"; shift
SCRIPT="$REPO/$1"; shift
export TMPDIR=/tmp
: > "$REPO/gpurun_out/${TAG}_pmc.txt"
for C in FETCH_SIZE WRITE_SIZE; do
  OUT=/tmp/pmc_${TAG}_$C
  rm -rf "$OUT"
  (cd /tmp && rocprofv3 --pmc $C --kernel-trace -d "$OUT" -o run -- python "$SCRIPT" "$@" > /dev/null 2> /tmp/pmc_$TAG.err) || { tail -20 /tmp/pmc_$TAG.err; exit 1; }
  DB=$(find "$OUT" -name "*.db" | head -1)
  python "$REPO/scripts/pmc_dump.py" "$DB" >> "$REPO/gpurun_out/${TAG}_pmc.txt"
done
cat "$REPO/gpurun_out/${TAG}_pmc.txt"
