#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_backbone_gpu.py tests/test_engine_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r02_t8.log
for i in 1 2; do WHICH=fwd,dgrad SHAPES=0,1,2 REPS=10 timeout 300 python scratch/bench_kernels.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r02_k8.log
WHICH=epi SHAPES=0,1,2 REPS=10 timeout 300 python scratch/bench_kernels.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_epi8.log
NBDT_HIP_LIB=$PWD/scratch/variants/libnbdt_tim.so timeout 300 python scratch/pp_timing.py 2>&1 | grep -v amdgpu.ids | head -6 | tee gpurun_out/r02_timing8.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --agreement-n 0 2>&1 | tail -1 | tee gpurun_out/r02_b8.log
