#!/bin/bash
# Round 3, final evidence run: full GPU suite, smoke, headline bench, the other configs' throughput, traces of the final state.
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_final
mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 ) > $OUT/pytest.log 2>&1
tail -9 $OUT/pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $OUT/smoke.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_line.json; cut -c1-200 $OUT/bench_line.json
timeout 600 python scratch/bench_configs.py 2>/dev/null | grep "^{" > $OUT/configs.jsonl; cat $OUT/configs.jsonl | cut -c1-170
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-kernel-timer --agreement-n 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -o bench -- $B --steps 5 --warmup 2 --no-overlap > $OUT/trace1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace2 -o bench -- $B --steps 5 --warmup 2 --cu-share-force > $OUT/trace2.log 2>&1
python $R/scratch/kernel_stats_report.py $(find $OUT/trace1 -name "*kernel_stats.csv" | head -1) 7 $OUT/kernel_stats_one_stream.txt "final state, one stream" > /dev/null
python $R/scratch/kernel_stats_report.py $(find $OUT/trace2 -name "*kernel_stats.csv" | head -1) 7 $OUT/kernel_stats_two_streams.txt "final state, two streams, CU sharing forced" > /dev/null
python $R/scratch/step_timeline.py $(find $OUT/trace2 -name "*kernel_trace.csv" | head -1) > $OUT/step_timeline.txt; head -8 $OUT/step_timeline.txt
find $OUT -name "*kernel_trace.csv" -delete
