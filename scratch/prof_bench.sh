#!/bin/bash
# rocprofv3 passes over bench.py (run on the GPU box); summaries are copied into profiles/ afterwards
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r01}
OUT=$R/gpurun_out/prof_bench_$TAG
mkdir -p $OUT
cd /tmp
# per-kernel durations that must agree with bench.py's roofline pass: single stream (no concurrent kernel)
NBDT_NO_WGRAD_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer > $OUT/trace.log 2>&1
# the default (weight gradients on the second stream): kernel durations overlap
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_overlap -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer > $OUT/trace_overlap.log 2>&1
NBDT_NO_WGRAD_STREAM=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer > $OUT/fetch.log 2>&1
NBDT_NO_WGRAD_STREAM=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer > $OUT/write.log 2>&1
find $OUT -name "*.csv" -size +0 | head; ls -la $OUT/trace/* | head
