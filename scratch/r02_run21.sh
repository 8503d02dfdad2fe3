#!/bin/bash
# CU-sharing schedule: new parity tests, bench A/B (default = sharing on) against --no-cu-share, full bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_backbone_gpu.py tests/test_engine_gpu.py -x -q -s -k "cu_subset or cu_budget or cu_sharing or partial_row" 2>&1 | grep -v amdgpu.ids | tail -12
for i in 1 2; do
  for F in "--no-cu-share" ""; do
    echo "== bench.py ${F:-(default: CU sharing)}"
    timeout 300 python bench.py --no-cpu-baseline --agreement-n 0 --no-kernel-timer --steps 40 --warmup 5 $F 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('value','ms_per_step','cu_share')})"
  done
done
timeout 400 python bench.py 2>/dev/null | tail -1 > gpurun_out/r02b_bench_line.json
cat gpurun_out/r02b_bench_line.json
