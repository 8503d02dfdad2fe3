#!/bin/bash
S=${SHAPES:-0}
echo -n "base  "; WHICH=${WHICH:-fwd} SHAPES=$S REPS=10 python scratch/bench_kernels.py 2>&1 | tail -1
for D in "$@"; do
  echo -n "$D  "; NBDT_HIP_LIB=$PWD/scratch/variants/libnbdt_$D.so WHICH=${WHICH:-fwd} SHAPES=$S REPS=10 python scratch/bench_kernels.py 2>&1 | tail -1
done
