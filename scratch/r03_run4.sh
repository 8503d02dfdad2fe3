#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_run4
mkdir -p $OUT
( time timeout 900 python -m pytest tests/test_rules_gpu.py tests/test_engine_gpu.py tests/test_cabi.py -m gpu -x -q -k "fused_head or golden or cabi or deterministic" ) > $OUT/pytest_head.log 2>&1
tail -15 $OUT/pytest_head.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --agreement-n 0 2>/dev/null | tail -1 > $OUT/bench_line.json; cut -c1-200 $OUT/bench_line.json
timeout 600 python scratch/emul_parity.py 512 2>&1 | grep -v amdgpu.ids | tee $OUT/emul_parity.txt
