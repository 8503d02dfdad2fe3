#!/bin/bash
# First GPU call of the next round: re-validate and re-profile the split CU-sharing schedule (round 2 ran out of GPU
# minutes after adopting it).  Usage on the GPU box:  bash scratch/r03_first.sh   (about 6 minutes)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_first
mkdir -p $OUT
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
grep -E "passed|failed" $OUT/pytest.log | tail -2
# headline, exactly as the driver runs it, and the A/B against the schedules it replaced
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_line.json
cut -c1-400 $OUT/bench_line.json
timeout 200 python scratch/cu_share_ab.py --steps 30 --rounds 2 off 47:200:16:96 47:200:16:128:split190 2>&1 | grep -v amdgpu.ids | tee $OUT/cu_share_ab.txt
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-kernel-timer --agreement-n 0"
# one stream: a kernel's duration is its own (what bench.py's roofline pass must agree with)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $B --steps 5 --warmup 2 --no-overlap > $OUT/trace.log 2>&1
# default: two streams, CU sharing
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_overlap -o bench -- $B --steps 5 --warmup 2 > $OUT/trace_overlap.log 2>&1
python $R/scratch/kernel_stats_report.py $(find $OUT/trace -name "*kernel_stats.csv" | head -1) 7 $OUT/kernel_stats.txt "one stream, split schedule" | head -25
python $R/scratch/kernel_stats_report.py $(find $OUT/trace_overlap -name "*kernel_stats.csv" | head -1) 7 $OUT/kernel_stats_overlap.txt "two streams, split CU sharing" | head -25
python $R/scratch/cu_share_trace_report.py $(find $OUT/trace_overlap -name "*kernel_trace.csv" | head -1) $OUT/cu_share_trace.txt | tail -30
# HBM traffic of the dominant kernels (separate PMC passes; scratch/traffic_report.py)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o bench -- $B --steps 2 --warmup 1 --no-overlap > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o bench -- $B --steps 2 --warmup 1 --no-overlap > $OUT/write.log 2>&1
find $OUT -name "*.csv" -size +0 | wc -l
