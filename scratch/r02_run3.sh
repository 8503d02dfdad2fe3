#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for S in 0 1; do
SHAPES=$S WHICH=fwd timeout 300 bash scratch/ablate.sh s1 s2 s3 2>&1 | grep -v amdgpu.ids
SHAPES=$S WHICH=fwd timeout 300 bash scratch/ablate.sh s3 s2 s1 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r02_sched3.log
