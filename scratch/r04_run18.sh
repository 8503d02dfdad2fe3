#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run18; mkdir -p $OUT
{ python scratch/ab_engine_flag.py fwd_tiles_on_side --steps 30 --rounds 3 2>&1 | grep round
timeout 1200 python -m pytest tests/test_engine_gpu.py tests/test_models_gpu.py tests/test_main_gpu.py -x -q -m gpu 2>&1 | tail -3
} 2>&1 | grep -v amdgpu.ids | tee $OUT/log.txt
