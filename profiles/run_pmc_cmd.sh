#!/bin/bash
# usage (on the GPU box, from the repo root): bash profiles/run_pmc_cmd.sh <tag> <python script> [args...]
# The two rocprofv3 --pmc passes of run_pmc.sh (FETCH_SIZE, WRITE_SIZE: gfx950 cannot hold both in one pass) around any python
# script of this repository (e.g. profiles/lde_probe.py 136 21), combined by pmc_summary.py into per-dispatch HBM bytes.
set -e
TAG=$1; shift
ROOT=$(pwd)
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  OUT=$ROOT/gpurun_out/pmc_${TAG}_$C
  mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -- python3 $ROOT/$1 "${@:2}" > /dev/null 2> $ROOT/gpurun_out/pmc_${TAG}_$C.err
done
cd $ROOT
F=$(find gpurun_out/pmc_${TAG}_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find gpurun_out/pmc_${TAG}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 profiles/pmc_summary.py $F $W gpurun_out/${TAG}_pmc_traffic.json > gpurun_out/${TAG}_pmc_traffic.txt
find gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE -name "*.csv" -size +3M -delete
