#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_rules_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/rules_test.log
timeout 300 python scratch/bench_rules.py > gpurun_out/rules_bench.jsonl 2> gpurun_out/rules_bench.err
cat gpurun_out/rules_test.log; cat gpurun_out/rules_bench.jsonl; tail -5 gpurun_out/rules_bench.err
