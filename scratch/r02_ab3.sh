#!/bin/bash
# in-step per-kernel durations for two libraries: kernel-trace of bench.py --no-overlap
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for L in intree $1; do
  OUT=$R/gpurun_out/ab3_$L; mkdir -p $OUT
  if [ $L = intree ]; then unset NBDT_HIP_LIB; else export NBDT_HIP_LIB=$R/scratch/variants/libnbdt_$L.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python $R/bench.py --no-cpu-baseline --no-kernel-timer --agreement-n 0 --steps 5 --warmup 2 --no-overlap > $OUT/log.txt 2>&1
  echo "== $L"; tail -1 $OUT/log.txt | cut -c1-120
  python - <<PY
import csv,glob
f=glob.glob('$OUT/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print(f"{float(r['TotalDurationNs'])/7e3:9.1f} us/step  n {int(r['Calls'])/7:5.1f}  avg {float(r['AverageNs'])/1e3:7.1f} us  {r['Name'][:70]}")
PY
done
