#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run14; mkdir -p $OUT
{ timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_models_gpu.py tests/test_baseline_configs_gpu.py -q -m gpu 2>&1 | tail -6
for S in "" "--no-cu-share"; do
python bench.py --no-cpu-baseline --no-kernel-timer --agreement-n 0 --no-other-configs --steps 20 --warmup 5 $S | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench $S', d['value'], d['ms_per_step'], d.get('cu_share'))"
done
python scratch/cu_share_ab.py --steps 30 --rounds 2 off 47:200:16:128 47:200:16:128:split190 2>&1 | tail -8
} 2>&1 | grep -v amdgpu.ids | tee $OUT/log.txt
