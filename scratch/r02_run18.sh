#!/bin/bash
# float4 / 128-row-lane partial-row folds: parity tests, step time A/B against the previous fold, kernel durations
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_backbone_gpu.py -x -q -k "fold or statistics or backward_sums or batchnorm or bench_shape" 2>&1 | tail -5
for i in 1 2 3; do
  for L in scratch/variants/libnbdt_oldfold.so ""; do
    echo "== lib ${L:-in-tree}"
    NBDT_ALLOW_TIMING_BUILD=1 NBDT_HIP_LIB=$L timeout 300 python bench.py --no-cpu-baseline --agreement-n 0 --no-kernel-timer --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')})"
  done
done
export TMPDIR=/tmp; R=$PWD; cd /tmp
for V in oldfold base; do
  if [ "$V" = base ]; then L=""; else L="NBDT_ALLOW_TIMING_BUILD=1 NBDT_HIP_LIB=$R/scratch/variants/libnbdt_$V.so"; fi
  env $L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$V -o b -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer --agreement-n 0 --no-overlap > /tmp/ks_$V.log 2>&1
  echo "== $V"; python - <<PY
import csv,glob
f=glob.glob('/tmp/ks_$V/**/*kernel_stats.csv',recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    tot+=float(r['TotalDurationNs'])
    if 'fold' in r['Name'] or 'finalize' in r['Name']:
        print(f"{float(r['AverageNs'])/1e3:8.1f} us x{int(r['Calls'])/7:5.1f}  min {float(r['MinNs'])/1e3:6.1f} max {float(r['MaxNs'])/1e3:6.1f}  {r['Name'][:60]}")
print('sum of kernel time per step (ms):', tot/7e6)
PY
  cp $(ls /tmp/ks_$V/*/*kernel_stats.csv | head -1) $R/gpurun_out/ks_$V.csv
done
