#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_backbone_gpu.py -x -q -m gpu -k "pingpong or bench_shape" > gpurun_out/r02_t3.log 2>&1
echo "pytest conv rc=$?" >> gpurun_out/r02_t3.log
tail -3 gpurun_out/r02_t3.log
for i in 1 2; do
WHICH=fwd,dgrad SHAPES=0,1,2 REPS=10 timeout 300 python scratch/bench_kernels.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r02_k2.log
SHAPES=0 WHICH=fwd timeout 300 bash scratch/ablate.sh pp_noepi pp_noa pp_nodma 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_abl2.log
WHICH=epi SHAPES=0,1,2 REPS=10 timeout 300 python scratch/bench_kernels.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_epi2.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --agreement-n 0 2>&1 | tail -1 | tee gpurun_out/r02_b2.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --agreement-n 0 --no-kernel-timer --no-overlap 2>&1 | tail -1 | tee gpurun_out/r02_b2_nostream.log
