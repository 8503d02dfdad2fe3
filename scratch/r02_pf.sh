#!/bin/bash
# quick check of an igemm change: backbone parity tests, kernel microbench, bench line
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_backbone_gpu.py tests/test_engine_gpu.py -x -q 2>&1 | tail -3
WHICH=fwd,dgrad SHAPES=0,1,2 REPS=10 python scratch/bench_kernels.py 2>&1 | grep -v amdgpu.ids
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['achieved'], d['roofline']['frac'])"
