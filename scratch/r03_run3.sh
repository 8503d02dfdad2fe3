#!/bin/bash
# Round 3, GPU call 3: deterministic mode tests + default-mode regression check (bench, cu_share A/B)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_run3
mkdir -p $OUT
( time timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -x -q --durations=8 -k "deterministic or fp32_oracle or self_consistent" -s ) > $OUT/pytest_det.log 2>&1
tail -25 $OUT/pytest_det.log
grep "split vs default" $OUT/pytest_det.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_line.json; cut -c1-200 $OUT/bench_line.json
timeout 600 python -m pytest tests/test_backbone_gpu.py tests/test_models_gpu.py -m gpu -x -q > $OUT/pytest_bb.log 2>&1; tail -3 $OUT/pytest_bb.log
