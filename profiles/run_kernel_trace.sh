#!/bin/bash
# usage (on the GPU box, from the repo root): bash profiles/run_kernel_trace.sh <tag> [bench args...]
# rocprofv3 kernel trace of bench.py; the per-(kernel, grid) summary lands in gpurun_out/<tag>_kernels.txt
set -e
TAG=$1; shift
ROOT=$(pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --no-cpu-baseline --no-variants --steps 4 --warmup 1 "$@" > $ROOT/gpurun_out/${TAG}_bench.json 2> $ROOT/gpurun_out/${TAG}_bench.err
cd $ROOT
TRACE=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 profiles/summarize_trace.py $TRACE > gpurun_out/${TAG}_kernels.txt
STATS=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp $STATS gpurun_out/${TAG}_kernel_stats.csv
find $OUT -name "*.csv" -size +2M -delete
head -40 gpurun_out/${TAG}_kernels.txt
