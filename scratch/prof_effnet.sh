#!/bin/bash
# usage: scratch/prof_effnet.sh TAG  -> gpurun_out/TAG_effnet_stats.csv
TAG=${1:-eff}
R=$PWD
python scratch/bench_effnet.py --steps 10 > gpurun_out/${TAG}_effnet_bench.json 2>gpurun_out/${TAG}_effnet_err.log
cat gpurun_out/${TAG}_effnet_bench.json
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o eff -- python $R/scratch/bench_effnet.py --steps 4 --warmup 2 > /tmp/prof_$TAG.log 2>&1
F=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1)
cp "$F" $R/gpurun_out/${TAG}_effnet_kernel_stats.csv
head -40 "$F" | cut -c1-200
