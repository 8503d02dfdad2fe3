#!/bin/bash
# Round 4 call 3: full kernel timeline of one step in the shipped two-stream schedule
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run3; mkdir -p $OUT
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-kernel-timer --agreement-n 0 --no-other-configs"
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace2 -o bench -- $B --steps 4 --warmup 2 > $OUT/trace2.log 2>&1
T=$(find $OUT/trace2 -name "*kernel_trace.csv" | head -1)
python $R/scratch/step_dump.py $T > $OUT/step_dump.txt
python $R/scratch/step_timeline.py $T > $OUT/step_timeline.txt; head -24 $OUT/step_timeline.txt
find $OUT -name "*kernel_trace.csv" -delete
