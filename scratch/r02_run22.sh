#!/bin/bash
# round-2b final: whole GPU suite, kernel trace of the default (CU-sharing, two streams) bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 600 python -m pytest tests -m gpu -x -q ) > gpurun_out/r22_pytest.log 2>&1
grep -E "passed|failed" gpurun_out/r22_pytest.log | tail -3
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o b -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer --agreement-n 0 > /tmp/ks.log 2>&1
grep -o '"value": [0-9.]*, "unit": "images/sec"' /tmp/ks.log
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r22_kernel_stats_overlap.csv
python $R/scratch/cu_share_trace_report.py $(find /tmp/ks -name "*kernel_trace.csv" | head -1) $R/gpurun_out/r22_cu_share_trace.txt | tail -50
