#!/bin/bash
# per-kernel average durations of the serial (single-stream) training step, for A/B of kernel variants
export TMPDIR=/tmp; R=$PWD; cd /tmp
for V in "$@"; do
  if [ "$V" = base ]; then L=""; else L="NBDT_ALLOW_TIMING_BUILD=1 NBDT_HIP_LIB=$R/scratch/variants/libnbdt_$V.so"; fi
  env $L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$V -o b -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-other-configs --agreement-n 0 --no-overlap > /tmp/ks_$V.log 2>&1
  echo "== $V"; python - <<PY
import csv,glob
f=glob.glob('/tmp/ks_$V/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'halo' in r['Name'] or 'igemm_dma' in r['Name']:
        print(f"{float(r['AverageNs'])/1e3:8.1f} us x{int(r['Calls'])//7:3d}  {r['Name'][:70]}")
PY
done
