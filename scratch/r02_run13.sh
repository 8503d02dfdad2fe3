#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_backbone_gpu.py -x -q -m gpu -k "weight_gradient or bench_shape" 2>&1 | grep -v "^$" | tail -40 | tee gpurun_out/r02_t13.log
