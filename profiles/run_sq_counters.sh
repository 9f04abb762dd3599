#!/bin/bash
# usage: bash profiles/run_sq_counters.sh <tag> <kernel-substring> -- <python script + args>
# SQ wave-state / instruction-mix counters per kernel (rocprofv3 --pmc, kernel trace only), summed over dispatches of kernels that match.
set -e
TAG=$1; PAT=$2; shift 3
ROOT=$(pwd); export TMPDIR=/tmp; cd /tmp
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAVES"; do
  N=$(echo $SET | awk '{print $1}')
  OUT=$ROOT/gpurun_out/sq_${TAG}_$N; mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT -- python3 $ROOT/"$@" > /dev/null 2> $ROOT/gpurun_out/sq_${TAG}_$N.err || true
  F=$(find $OUT -name "*counter_collection.csv" | head -1)
  python3 - "$F" "$PAT" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in acc: print("%-24s %16.0f  (%d dispatches)" % (k, acc[k], n[k]))
PY
  find $OUT -name "*.csv" -size +2M -delete
done
