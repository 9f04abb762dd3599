#!/bin/bash
# usage (GPU box, repo root): bash profiles/pow_ab.sh <tag>     -- k_pow_batch2 duration and VALU wave-instructions per dispatch (rocprofv3 --pmc SQ_INSTS_VALU)
set -e
TAG=$1
ROOT=$(pwd); export TMPDIR=/tmp; cd /tmp
OUT=$ROOT/gpurun_out/powab_$TAG; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d $OUT -- python3 $ROOT/profiles/zkdsa_batch_trace.py > $ROOT/gpurun_out/powab_$TAG.out 2> $ROOT/gpurun_out/powab_$TAG.err || true
python3 - $(find $OUT -name "*counter_collection.csv" | head -1) $(find $OUT -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_pow_batch2" in r["Kernel_Name"]:
        print("%s  SQ_INSTS_VALU %.4g  -> %.2f M candidates at 12.8 k VALU wave-instructions per 64" % (r["Kernel_Name"][:24], float(r["Counter_Value"]), float(r["Counter_Value"]) / 12800 * 64 / 1e6))
for r in csv.DictReader(open(sys.argv[2])):
    if "k_pow_batch2" in r["Kernel_Name"]:
        print("%s  %.3f ms" % (r["Kernel_Name"][:24], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
grep candidates $ROOT/gpurun_out/powab_$TAG.out
find $OUT -name "*.csv" -size +2M -delete
