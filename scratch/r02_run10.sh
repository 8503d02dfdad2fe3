#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r02b
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-overlap --agreement-n 0 > $OUT/trace.log 2>&1
ls $OUT/trace | head
python - <<PY
import csv, glob, collections
f = glob.glob('$OUT/trace/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
for r in rows[:28]:
    print(f"{float(r['TotalDurationNs'])/7e3:9.1f} us/step  calls/step {int(r['Calls'])/7:5.1f}  avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.2f}%  {r['Name'][:110]}")
PY
