#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run4; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_backbone_gpu.py -x -q -m gpu -k "batchnorm or fold or statistics" 2>&1 | tail -5
  timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "deterministic or fused_head or self_consistent" 2>&1 | tail -5
  python scratch/ab_engine_flag.py fuse_bn_fold --steps 30 --rounds 3 2>&1 | grep round ) | tee $OUT/log.txt
