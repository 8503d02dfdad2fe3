#!/bin/bash
# rocprofv3 passes over bench.py (run on the GPU box); summaries are copied into profiles/ afterwards.
# PMC counters are collected in their own passes (never together with --kernel-trace).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
OUT=$R/gpurun_out/prof_bench_$TAG
mkdir -p $OUT
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-kernel-timer --agreement-n 0 --no-other-configs --cu-share-force"
# per-kernel durations that must agree with bench.py's roofline pass: single stream (no concurrent kernel)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $B --steps 5 --warmup 2 --no-overlap > $OUT/trace.log 2>&1
# the default (weight gradients on the second stream): kernel durations overlap
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_overlap -o bench -- $B --steps 5 --warmup 2 > $OUT/trace_overlap.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o bench -- $B --steps 2 --warmup 1 --no-overlap > $OUT/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o bench -- $B --steps 2 --warmup 1 --no-overlap > $OUT/write.log 2>&1
# SQ / LDS / clock counters of the final kernels (MFMA utilisation, wait states)
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq1 -o bench -- $B --steps 2 --warmup 1 --no-overlap > $OUT/sq1.log 2>&1
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $OUT/sq2 -o bench -- $B --steps 2 --warmup 1 --no-overlap > $OUT/sq2.log 2>&1
timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/clk -o bench -- $B --steps 2 --warmup 1 --no-overlap > $OUT/clk.log 2>&1
find $OUT -name "*.csv" -size +0 | wc -l
