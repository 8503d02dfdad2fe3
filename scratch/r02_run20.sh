#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python scratch/cu_share_ab.py --steps 30 --rounds 2 off 47:230:16:96:noprio 47:230:16:96:noprio-nojoin 47:200:16:96:noprio 47:200:16:96:noprio-nojoin 47:180:16:128:noprio 47:180:16:128:noprio-nojoin 47:160:16:128:noprio 47:160:16:128:noprio-nojoin 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r20_cu_share_ab.txt
