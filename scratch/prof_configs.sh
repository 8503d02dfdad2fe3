#!/bin/bash
# kernel statistics of the BASELINE configurations named on the command line (c1 c3 c4 c4inf c5), one stream
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_configs; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for C in "$@"; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_$C -o t -- python $R/scratch/run_config.py $C --steps 5 --warmup 2 --one-stream > $OUT/tr_$C.log 2>&1
  python $R/scratch/kernel_stats_report.py $(find $OUT/tr_$C -name "*kernel_stats.csv" | head -1) 7 $OUT/${C}_kernel_stats_one_stream.txt "$C, one stream, 7 steps profiled (2 warm-up + 5)" > /dev/null
  tail -2 $OUT/tr_$C.log
done
find $OUT -name "*kernel_trace.csv" -delete
