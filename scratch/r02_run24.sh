#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 80 python scratch/cu_share_ab.py --steps 25 --rounds 2 off 47:200:16:128:split190 47:200:16:160:split190 47:200:16:128:split175 47:200:24:128:split205 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r24_cu_share_split3.txt
