#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run9; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/scratch/variants
{ echo "# strided 3x3 conv (32x32x160 -> 16x16x320), first-generation kernels: base vs contiguous weight pieces (timing only)"
WHICH=fwd,dgrad SHAPES=3 python scratch/bench_kernels.py
NBDT_ALLOW_TIMING_BUILD=1 NBDT_HIP_LIB=$V/libnbdt_wtfake.so WHICH=fwd,dgrad SHAPES=3 python scratch/bench_kernels.py
bash scratch/ab_bench.sh "X=0" "NBDT_ALLOW_TIMING_BUILD=1 NBDT_HIP_LIB=$V/libnbdt_wtfake.so"
} 2>&1 | grep -v "amdgpu.ids" | tee $OUT/log.txt
