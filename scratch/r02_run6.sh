#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_backbone_gpu.py -x -q -m gpu -k "pingpong or bench_shape" 2>&1 | tail -3 | tee gpurun_out/r02_t6.log
for i in 1 2; do WHICH=fwd,dgrad SHAPES=0,1,2 REPS=10 timeout 300 python scratch/bench_kernels.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r02_k6.log
SHAPES=0 WHICH=fwd timeout 300 bash scratch/ablate.sh pp_noepi s1 s2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_abl6.log
NBDT_HIP_LIB=$PWD/scratch/variants/libnbdt_tim.so timeout 300 python scratch/pp_timing.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_timing6.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --agreement-n 0 2>&1 | tail -1 | tee gpurun_out/r02_b6.log
