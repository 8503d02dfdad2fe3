#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_backbone_gpu.py tests/test_engine_gpu.py -x -q 2>&1 | tail -2
V=scratch/variants/libnbdt_$1.so
for i in 1 2; do
for L in "" $V; do
  echo "== lib ${L:-in-tree}"
  NBDT_HIP_LIB=$L timeout 600 python bench.py --no-cpu-baseline --agreement-n 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['achieved'], d['roofline']['frac'])"
done
done
for L in "" $V; do
echo "== lib ${L:-in-tree}"
NBDT_HIP_LIB=$L WHICH=dgrad SHAPES=0,1,2 REPS=10 python scratch/bench_kernels.py 2>&1 | grep -v amdgpu.ids
done
