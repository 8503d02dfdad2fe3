#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_backbone_gpu.py -x -q -m gpu -k "pingpong or bench_shape or conv_forward" 2>&1 | tail -3 | tee gpurun_out/r02_t9.log
for F in 2 3 2 3; do echo "FORCE=$F"; FORCE=$F WHICH=fwd,dgrad SHAPES=0,1,2 REPS=10 timeout 300 python scratch/bench_kernels.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r02_k9.log
for F in 2 3; do echo "FORCE=$F"; FORCE=$F WHICH=epi SHAPES=0,1,2 REPS=10 timeout 300 python scratch/bench_kernels.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r02_epi9.log
