#!/bin/bash
cd $GRAFT_REPO_ROOT
for S in 0 2; do
SHAPES=$S WHICH=fwd timeout 300 bash scratch/ablate.sh s2 s4 s6 s7 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r02_sched7.log
