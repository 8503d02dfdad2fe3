#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_backbone_gpu.py -x -q -m gpu -k "weight_gradient or bench_shape or conv_forward" 2>&1 | tail -5 | tee gpurun_out/r02_t12.log
for V in 2 3 2 3; do echo "VARIANT=$V"; VARIANT=$V WHICH=wgrad SHAPES=0,1,2 REPS=10 timeout 300 python scratch/bench_kernels.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r02_k12.log
NBDT_HIP_LIB=$PWD/scratch/variants/libnbdt_wtim.so timeout 300 python scratch/wpp_timing.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_wtiming.log
