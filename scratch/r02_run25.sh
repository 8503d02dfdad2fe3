#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 70 python bench.py --no-cpu-baseline --agreement-n 0 --steps 20 --warmup 5 2>&1 | grep -v amdgpu.ids | tail -3 > gpurun_out/r02c_bench_line.json
cat gpurun_out/r02c_bench_line.json | cut -c1-2600
