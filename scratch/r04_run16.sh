#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run16; mkdir -p $OUT
{ timeout 1200 python -m pytest tests/test_backbone_gpu.py -x -q -m gpu -k "fold or statistics or batchnorm or forward_backward" 2>&1 | tail -3
for i in 1 2; do python bench.py --no-cpu-baseline --no-kernel-timer --agreement-n 0 --no-other-configs --steps 20 --warmup 5 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])"; done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/log.txt
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr -o b -- python $R/bench.py --no-cpu-baseline --no-kernel-timer --agreement-n 0 --no-other-configs --steps 5 --warmup 2 --no-overlap > $OUT/tr.log 2>&1
python $R/scratch/kernel_stats_report.py $(find $OUT/tr -name "*kernel_stats.csv" | head -1) 7 $OUT/kernel_stats_one_stream.txt "after the statistics fold, one stream" | head -30
find $OUT -name "*kernel_trace.csv" -delete
