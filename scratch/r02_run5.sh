#!/bin/bash
cd $GRAFT_REPO_ROOT
NBDT_HIP_LIB=$PWD/scratch/variants/libnbdt_tim.so timeout 300 python scratch/pp_timing.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_timing7.log
