#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run8; mkdir -p $OUT
( time timeout 1700 python -m pytest tests/test_baseline_configs_gpu.py -m gpu -q -s -k "config5" ) > $OUT/pytest.log 2>&1
grep -n "config 5\|passed\|failed\|^E  \|real" $OUT/pytest.log | cut -c1-300 | tail -20
