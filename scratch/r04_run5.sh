#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_run5; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_analysis_gpu.py tests/test_main_gpu.py tests/test_bench_gpu.py -x -q -m gpu 2>&1 | tail -15
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>$OUT/bench_err.txt | tail -1 > $OUT/bench_line.json; python -c "
import json; d=json.load(open('$OUT/bench_line.json')); print(d['value'], d['ms_per_step'], d['roofline']['traffic'], d['roofline']['traffic_source'][:90]); print(d['agreement'])
for o in d['other_configs']: print(o['config'][:40], o['value'], o['ms_per_step'], o['bound'], o['frac'])
print(d['cpu_baseline']['value'])" ) 2>&1 | tee $OUT/log.txt
