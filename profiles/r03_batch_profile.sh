set -e
mkdir -p gpurun_out
B="python bench.py --workload zkdsa-batch --steps 8 --warmup 3 --no-cpu-baseline"
for cfg in "t1 --threads 1 --sub-batch 256" "t2 --threads 2 --sub-batch 128" "t4 --threads 4 --sub-batch 64"; do
  set -- $cfg; tag=$1; shift
  $B "$@" 2>/dev/null > gpurun_out/r03b_zkdsa_$tag.json
done
$B --batch 2048 --threads 4 --sub-batch 256 2>/dev/null > gpurun_out/r03b_zkdsa_2048.json
python bench.py --workload smt --steps 3 --no-cpu-baseline 2>/dev/null > gpurun_out/r03b_smt.json
python bench.py --workload keccak256 --steps 3 --no-cpu-baseline 2>/dev/null > gpurun_out/r03b_keccak.json
GLP_BATCH_TRACE=1 python profiles/zkdsa_batch_trace.py > gpurun_out/r03b_trace.txt 2>&1
bash profiles/pow_ab.sh final > gpurun_out/r03b_pow.txt 2>&1
export TMPDIR=/tmp; R=$(pwd); mkdir -p gpurun_out/tl2 && cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl2 -- python3 $R/profiles/batch_timeline.py > $R/gpurun_out/tl2.out 2>&1; cd $R
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03b_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], 'value %.0f %s  ms_per_step %.3f  verified %s  prove+verify %s' % (d['value'], d['unit'], d['ms_per_step'], d.get('verified'), json.dumps(d.get('prove_plus_verify'))[:120]))
    except Exception as e: print(f, 'ERR', e)
PY
for W in zkdsa smt keccak; do
  mkdir -p gpurun_out/tlf_$W && cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tlf_$W -- python3 $R/profiles/batch_timeline.py $W > $R/gpurun_out/tlf_$W.out 2>&1; cd $R
  python3 profiles/batch_timeline.py summarize $(find gpurun_out/tlf_$W -name "*kernel_trace.csv" | head -1) | cut -c1-150 > gpurun_out/r03b_timeline_$W.txt
  tail -1 gpurun_out/r03b_timeline_$W.txt
done
