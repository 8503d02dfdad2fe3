#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_backbone_gpu.py tests/test_engine_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r02_t16.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --agreement-n 0 2>&1 | tail -1 | tee gpurun_out/r02_b16.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --agreement-n 0 --no-overlap --no-kernel-timer 2>&1 | tail -1 | tee gpurun_out/r02_b16_noov.log
