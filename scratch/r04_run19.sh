#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run19; mkdir -p $OUT
{ timeout 1200 python -m pytest tests/test_backbone_gpu.py tests/test_effnet_gpu.py tests/test_models_gpu.py -x -q -m gpu 2>&1 | tail -3
for C in c4 c1 c5; do python scratch/run_config.py $C --steps 20 2>/dev/null | tail -1; done
python bench.py --no-cpu-baseline --agreement-n 0 --no-other-configs --steps 20 --warmup 5 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['traffic'], d['roofline']['traffic_source'])"
} 2>&1 | grep -v amdgpu.ids | tee $OUT/log.txt
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr -o b -- python $R/scratch/run_config.py c4 --steps 5 --warmup 2 --one-stream > $OUT/tr.log 2>&1
python $R/scratch/kernel_stats_report.py $(find $OUT/tr -name "*kernel_stats.csv" | head -1) 7 $OUT/c4_kernel_stats_one_stream.txt "c4 one stream" | grep "stem\|sum of"
find $OUT -name "*kernel_trace.csv" -delete
