#!/bin/bash
# full check: GPU test suite, profiling passes, default bench line
cd $GRAFT_REPO_ROOT
TAG=${1:-r02}
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
tail -4 gpurun_out/${TAG}_pytest.log
bash scratch/prof_bench.sh $TAG
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_err.log
tail -1 gpurun_out/${TAG}_bench_line.json
