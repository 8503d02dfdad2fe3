#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 100 python scratch/cu_share_ab.py --steps 30 --rounds 2 off 47:200:16:96 47:200:16:96:fit 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r23_cu_share_fit.txt
