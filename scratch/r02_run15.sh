#!/bin/bash
cd $GRAFT_REPO_ROOT
NBDT_HIP_LIB=$PWD/scratch/variants/libnbdt_wtim3.so timeout 300 python scratch/wpp_timing.py 2>&1 | grep -v amdgpu.ids | head -2 | tee gpurun_out/r02_wtiming2.log
