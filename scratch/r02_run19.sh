#!/bin/bash
# bench-configuration parity test (WRN-28-10, B=512, fp32 CPU oracle) and the CU-sharing probe
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
free -g | head -2; nproc
( time timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -s -k "bench_configuration" ) > gpurun_out/r19_parity.log 2>&1
tail -5 gpurun_out/r19_parity.log
grep -E "logit err|worst gradient" gpurun_out/r19_parity.log
timeout 120 ./probes/cu_share_probe > gpurun_out/r19_cu_share.txt 2>&1
cat gpurun_out/r19_cu_share.txt
