#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 60 python -m pytest tests/test_engine_gpu.py -x -q -s -k "cu_sharing" 2>&1 | grep -v amdgpu.ids | tail -5
timeout 60 python bench.py --no-cpu-baseline --agreement-n 0 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r02c_bench_line.json
cat gpurun_out/r02c_bench_line.json | cut -c1-900
