#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run6; mkdir -p $OUT
( time timeout 1700 python -m pytest tests -m gpu -q --durations=8 ) > $OUT/pytest.log 2>&1
tail -16 $OUT/pytest.log
