#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_run6
mkdir -p $OUT
timeout 500 python scratch/cu_share_ab.py --steps 30 --rounds 2 47:200:16:128:split190 47:200:16:128:split150 47:200:16:160:split130 47:200:16:128:split230 47:200:16:128:split270 47:200:16:96:split190 2>&1 | grep -v amdgpu.ids | tee $OUT/split_sweep.txt
