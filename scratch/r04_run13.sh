#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run13; mkdir -p $OUT
{ timeout 1200 python -m pytest tests/test_backbone_gpu.py tests/test_effnet_gpu.py -x -q -m gpu 2>&1 | tail -4
WHICH=epi SHAPES=0,1,2 python scratch/bench_kernels.py
NBDT_HIP_LIB=$PWD/scratch/variants/libnbdt_tim.so STATS=1 python scratch/pp_timing.py | grep -A2 "B=512 H=32" | grep epilogue
python bench.py --no-cpu-baseline --no-kernel-timer --agreement-n 0 --no-other-configs --steps 20 --warmup 5 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])"
python bench.py --no-cpu-baseline --no-kernel-timer --agreement-n 0 --no-other-configs --steps 20 --warmup 5 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])"
} 2>&1 | grep -v amdgpu.ids | tee $OUT/log.txt
