#!/bin/bash
# Round 6, final evidence run: full GPU suite, smoke, the driver's bench command, rocprofv3 passes of the final state
# (kernel stats one / two streams, FETCH / WRITE, SQ counters), the step timeline, the other configurations' profiles.
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/${R06_OUT:-r06_final}
mkdir -p $OUT
if [ "$1" != "noprof" ]; then
( time timeout 1500 python -m pytest tests -m gpu -q --durations=5 ) > $OUT/pytest.log 2>&1
tail -9 $OUT/pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $OUT/smoke.txt
fi
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2>$OUT/bench_err.txt | tail -1 > $OUT/bench_line.json; cut -c1-200 $OUT/bench_line.json
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
bash $R/scratch/prof_bench.sh r06 > $OUT/prof_bench.log 2>&1
P=$R/gpurun_out/prof_bench_r06
python $R/scratch/kernel_stats_report.py $(find $P/trace -name "*kernel_stats.csv" | head -1) 7 $OUT/kernel_stats_one_stream.txt "final state of round 6, one stream (bench.py --no-overlap --cu-share-force)" > /dev/null
python $R/scratch/kernel_stats_report.py $(find $P/trace_overlap -name "*kernel_stats.csv" | head -1) 7 $OUT/kernel_stats_two_streams.txt "final state of round 6, two streams, CU sharing forced" > /dev/null
python $R/scratch/step_timeline.py $(find $P/trace_overlap -name "*kernel_trace.csv" | head -1) > $OUT/step_timeline.txt; head -8 $OUT/step_timeline.txt
python $R/scratch/step_dump.py $(find $P/trace_overlap -name "*kernel_trace.csv" | head -1) > $OUT/step_dump.txt
python $R/scratch/traffic_report.py $P $OUT/final 3 > /dev/null
python $R/scratch/pmc_bench_report.py $P $OUT/final > /dev/null 2>&1
cd /tmp
for C in c1 c3 c4 c4inf c5; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_$C -o t -- python $R/scratch/run_config.py $C --steps 5 --warmup 2 --one-stream > $OUT/tr_$C.log 2>&1
  python $R/scratch/kernel_stats_report.py $(find $OUT/tr_$C -name "*kernel_stats.csv" | head -1) 7 $OUT/${C}_kernel_stats_one_stream.txt "$C, one stream, 7 steps profiled (2 warm-up + 5)" > /dev/null
done
# EfficientNet-B0 (config 5): HBM-side bytes of one training step, by kernel family (the LAST of two steps: the first one
# zero-fills every buffer it allocates, +3 GB of writes)
bash $R/scratch/c5_traffic_by_kernel.sh ${R06_OUT:-r06_final} > /dev/null 2>&1; head -4 $OUT/c5_traffic_by_kernel.txt
find $OUT $P -name "*kernel_trace.csv" -delete; find $OUT $P -name "*counter_collection.csv" -size +20M -delete
du -sh $OUT $P
