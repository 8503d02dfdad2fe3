#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run7; mkdir -p $OUT
( time timeout 1700 python -m pytest tests/test_reference_fp32_gpu.py tests/test_backbone_gpu.py -m gpu -q -s -k "fp32 or reserved" ) > $OUT/pytest.log 2>&1
grep -v "^$" $OUT/pytest.log | tail -40
