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
# EfficientNet-B0 (config 5): HBM-side bytes of one training step
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c5_fetch -o t -- python $R/scratch/run_config.py c5 --steps 1 --warmup 1 --one-stream > $OUT/c5_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/c5_write -o t -- python $R/scratch/run_config.py c5 --steps 1 --warmup 1 --one-stream > $OUT/c5_write.log 2>&1
python - <<PY
import csv, glob
def tot(d, name):
    s = 0.0
    for p in glob.glob(f"$OUT/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] == name: s += float(r["Counter_Value"])
    return s * 1024.0
f, w = tot("c5_fetch", "FETCH_SIZE"), tot("c5_write", "WRITE_SIZE")
open("$OUT/c5_hbm_traffic.txt", "w").write(
    "# EfficientNet-B0 / Imagenet1000 224x224, batch 128: HBM-side bytes of TWO training steps (1 warm-up + 1), one stream,\n"
    "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (KB as reported; FETCH x2 = gfx950 correction)\n"
    f"FETCH_SIZE {f/1e9:.2f} GB raw, {2*f/1e9:.2f} GB corrected; WRITE_SIZE {w/1e9:.2f} GB; per step {(2*f+w)/2e9:.2f} GB\n")
print(open("$OUT/c5_hbm_traffic.txt").read())
PY
find $OUT $P -name "*kernel_trace.csv" -delete; find $OUT $P -name "*counter_collection.csv" -size +20M -delete
du -sh $OUT $P
