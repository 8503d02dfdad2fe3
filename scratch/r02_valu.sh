#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2; do
for L in "" scratch/variants/libnbdt_valu40.so scratch/variants/libnbdt_valu80.so; do
  echo "== lib ${L:-in-tree}"
  NBDT_HIP_LIB=$L WHICH=fwd SHAPES=0,1,2 REPS=10 python scratch/bench_kernels.py 2>&1 | grep -v amdgpu.ids
done
done
