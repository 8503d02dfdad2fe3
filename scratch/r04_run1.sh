#!/bin/bash
# Round 4, first call: this round's baseline on one box -- headline bench, the other configs' throughput and per-kernel
# one-stream stats (rocprofv3 --kernel-trace --stats) of C1/C3/C4/C4inf/C5.
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run1
mkdir -p $OUT
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_line.json; cut -c1-300 $OUT/bench_line.json
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for C in c1 c3 c4 c4inf c5; do
  timeout 200 python $R/scratch/run_config.py $C --steps 10 2>/dev/null | tail -1 >> $OUT/configs.jsonl
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_$C -o t -- python $R/scratch/run_config.py $C --steps 5 --warmup 2 --one-stream > $OUT/tr_$C.log 2>&1
  python $R/scratch/kernel_stats_report.py $(find $OUT/tr_$C -name "*kernel_stats.csv" | head -1) 7 $OUT/${C}_kernel_stats_one_stream.txt "$C, one stream, 7 steps profiled (2 warm-up + 5)" > /dev/null
  find $OUT/tr_$C -name "*kernel_trace.csv" -delete
done
cat $OUT/configs.jsonl
