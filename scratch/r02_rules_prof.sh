#!/bin/bash
mkdir -p gpurun_out/rules_prof
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/rules_prof -o rules -- python /root/repo/scratch/bench_rules.py > /root/repo/gpurun_out/rules_prof/out.jsonl 2>/root/repo/gpurun_out/rules_prof/err.log
cd /root/repo
ls -R gpurun_out/rules_prof | head -20
