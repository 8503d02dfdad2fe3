#!/bin/bash
# Round 4 call 2: sensitivity of the step to the speed of the weight gradient / the igemm main loop (timing-only builds
# that skip part of the K loop), and what ds_read_b128 reads of a K-major gy image would buy the weight gradient.
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run2; mkdir -p $OUT
V=$GRAFT_REPO_ROOT/scratch/variants
{
echo "# wgrad microbench (bench_kernels.py): base vs gy read with ds_read_b128 (timing only)"
WHICH=wgrad SHAPES=0,1,2 python scratch/bench_kernels.py
NBDT_ALLOW_TIMING_BUILD=1 NBDT_HIP_LIB=$V/libnbdt_gyb128.so WHICH=wgrad SHAPES=0,1,2 python scratch/bench_kernels.py
NBDT_ALLOW_TIMING_BUILD=1 NBDT_HIP_LIB=$V/libnbdt_w7.so WHICH=wgrad SHAPES=0,1,2 python scratch/bench_kernels.py
echo "# step A/B: base / wgrad 7/8 / wgrad 6/8 / igemm K loop 4/5 (alternating, 2 rounds)"
bash scratch/ab_bench.sh "X=0" "NBDT_ALLOW_TIMING_BUILD=1 NBDT_HIP_LIB=$V/libnbdt_w7.so" "NBDT_ALLOW_TIMING_BUILD=1 NBDT_HIP_LIB=$V/libnbdt_w6.so" "NBDT_ALLOW_TIMING_BUILD=1 NBDT_HIP_LIB=$V/libnbdt_c45.so" "NBDT_ALLOW_TIMING_BUILD=1 NBDT_HIP_LIB=$V/libnbdt_gyb128.so"
} 2>&1 | grep -v Warning | tee $OUT/sensitivity.txt
