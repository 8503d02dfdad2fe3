#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_verify2
mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -q --durations=3 ) > $OUT/pytest.log 2>&1; grep "passed\|failed\|FAILED" $OUT/pytest.log | tail -4
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_line.json; python -c "
import json;d=json.load(open('$OUT/bench_line.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline_wgrad']['frac'],d['roofline_wgrad'].get('in_step_frac'),d['cu_share']['enabled'])"
