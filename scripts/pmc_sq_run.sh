#!/bin/bash
# SQ counters of one kernel (wave cycles, wait buckets, LDS conflicts, matrix-pipe busy) from rocprofv3, --pmc with
# --kernel-trace only (MI355X_MICROARCH.md: 8 SQ slots per pass).  Counter names this rocprofv3 does not list are dropped.
# usage (GPU box, repo root): bash scripts/pmc_sq_run.sh <tag> <python script> [args...]   -> gpurun_out/<tag>_sq_pmc.txt
REPO="$(cd "$(dirname "$0")/.." && pwd)"
TAG="$1"; shift
SCRIPT="$REPO/$1"; shift
export TMPDIR=/tmp
OUTF="$REPO/gpurun_out/${TAG}_sq_pmc.txt"
: > "$OUTF"
(cd /tmp && rocprofv3 -L > /tmp/pmc_list.txt 2>&1)
PASS1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
PASS2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM"
PASS3="SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES"
n=0
for P in "$PASS1" "$PASS2" "$PASS3"; do
  n=$((n+1))
  KEEP=""
  for C in $P; do
    if grep -qw "$C" /tmp/pmc_list.txt; then KEEP="$KEEP $C"; else echo "# not listed by rocprofv3 -L: $C" >> "$OUTF"; fi
  done
  [ -z "$KEEP" ] && continue
  OUT=/tmp/pmcsq_${TAG}_$n
  rm -rf "$OUT"
  (cd /tmp && timeout 600 rocprofv3 --pmc $KEEP --kernel-trace -d "$OUT" -o run -- python "$SCRIPT" "$@" > /dev/null 2> /tmp/pmcsq_$TAG.err) || { echo "# pass $n failed" >> "$OUTF"; tail -5 /tmp/pmcsq_$TAG.err >> "$OUTF"; continue; }
  DB=$(find "$OUT" -name "*.db" | head -1)
  python "$REPO/scripts/pmc_dump.py" "$DB" >> "$OUTF"
done
cat "$OUTF"
