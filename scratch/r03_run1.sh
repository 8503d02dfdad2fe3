#!/bin/bash
# Round 3, GPU call 1: profile the schedule bench.py times (split CU sharing) -- kernel traces (one / two streams),
# FETCH/WRITE traffic, SQ + clock counters in one-stream and in two-stream (confined-kernel) mode.
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_run1
mkdir -p $OUT
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_line.json
cut -c1-600 $OUT/bench_line.json
timeout 200 python scratch/cu_share_ab.py --steps 30 --rounds 2 off 47:200:16:96 47:200:16:128:split190 2>&1 | grep -v amdgpu.ids | tee $OUT/cu_share_ab.txt
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-kernel-timer --agreement-n 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $B --steps 5 --warmup 2 --no-overlap > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_overlap -o bench -- $B --steps 5 --warmup 2 > $OUT/trace_overlap.log 2>&1
python $R/scratch/kernel_stats_report.py $(find $OUT/trace -name "*kernel_stats.csv" | head -1) 7 $OUT/kernel_stats.txt "one stream, split schedule" | head -40
python $R/scratch/kernel_stats_report.py $(find $OUT/trace_overlap -name "*kernel_stats.csv" | head -1) 7 $OUT/kernel_stats_overlap.txt "two streams, split CU sharing" | head -40
python $R/scratch/cu_share_trace_report.py $(find $OUT/trace_overlap -name "*kernel_trace.csv" | head -1) $OUT/cu_share_trace.txt | tail -30
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o bench -- $B --steps 2 --warmup 1 --no-overlap > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o bench -- $B --steps 2 --warmup 1 --no-overlap > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq1 -o bench -- $B --steps 2 --warmup 1 --no-overlap > $OUT/sq1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/clk -o bench -- $B --steps 2 --warmup 1 --no-overlap > $OUT/clk.log 2>&1
# the same two counter passes with the second stream on: the CU-confined kernels and the CU-budgeted weight gradient
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $OUT/ov/sq1 -o bench -- $B --steps 2 --warmup 1 > $OUT/sq1_ov.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/ov/clk -o bench -- $B --steps 2 --warmup 1 > $OUT/clk_ov.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/ov/fetch -o bench -- $B --steps 2 --warmup 1 > $OUT/fetch_ov.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/ov/write -o bench -- $B --steps 2 --warmup 1 > $OUT/write_ov.log 2>&1
mkdir -p $OUT/ov/trace; cp $(find $OUT/trace_overlap -name "*kernel_stats.csv" | head -1) $OUT/ov/trace/
find $OUT -name "*.csv" -size +0 | wc -l
# keep what is merged back small: the per-dispatch traces are large
find $OUT -name "*kernel_trace.csv" -size +20M -delete
du -sh $OUT
