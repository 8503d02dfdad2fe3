#!/bin/bash
# two-stream kernel trace of bench.py -> step dump + timeline + kernel stats under gpurun_out/r06_trace_$1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-a}
OUT=$R/gpurun_out/r06_trace_$TAG
mkdir -p $OUT
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-kernel-timer --agreement-n 0 --no-other-configs --no-attainable --cu-share-force"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_overlap -o bench -- $B --steps 5 --warmup 2 > $OUT/trace_overlap.log 2>&1
tail -1 $OUT/trace_overlap.log | cut -c1-120
python $R/scratch/kernel_stats_report.py $(find $OUT/trace_overlap -name "*kernel_stats.csv" | head -1) 7 $OUT/kernel_stats_two_streams.txt "two streams, CU sharing forced ($TAG)" > /dev/null
python $R/scratch/step_timeline.py $(find $OUT/trace_overlap -name "*kernel_trace.csv" | head -1) > $OUT/step_timeline.txt; head -24 $OUT/step_timeline.txt
python $R/scratch/step_dump.py $(find $OUT/trace_overlap -name "*kernel_trace.csv" | head -1) > $OUT/step_dump.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete 2>/dev/null
du -sh $OUT
