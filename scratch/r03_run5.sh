#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_run5
mkdir -p $OUT
( timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "deterministic or fused_head or hipgraph or cu_sharing" ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
timeout 300 python scratch/ab_engine_flag.py debug_join_each_unit 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_join.txt
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-kernel-timer --agreement-n 0 --cu-share-force"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $B --steps 5 --warmup 2 > $OUT/trace.log 2>&1
python $R/scratch/step_timeline.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) | tee $OUT/step_timeline.txt
find $OUT -name "*kernel_trace.csv" -size +20M -delete
