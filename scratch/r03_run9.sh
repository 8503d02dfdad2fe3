#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_run9
mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 ) > $OUT/pytest.log 2>&1
tail -14 $OUT/pytest.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_line.json; cut -c1-260 $OUT/bench_line.json
