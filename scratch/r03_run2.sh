#!/bin/bash
# Round 3, GPU call 2: the new tests, the CU-mask / contention probes, PMC of the confined kernels (forced schedule).
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_run2
mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 ) > $OUT/pytest.log 2>&1
tail -30 $OUT/pytest.log
timeout 120 ./probes/cu_mask_probe > $OUT/cu_mask_probe.txt 2>&1; tail -40 $OUT/cu_mask_probe.txt
timeout 300 python scratch/contention_probe.py > $OUT/contention.txt 2>&1; grep -v amdgpu.ids $OUT/contention.txt | tail -8
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_line.json; cut -c1-300 $OUT/bench_line.json
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-kernel-timer --agreement-n 0 --cu-share-force"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $OUT/ov/sq1 -o bench -- $B --steps 2 --warmup 1 > $OUT/sq1_ov.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/ov/clk -o bench -- $B --steps 2 --warmup 1 > $OUT/clk_ov.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/ov/fetch -o bench -- $B --steps 2 --warmup 1 > $OUT/fetch_ov.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/ov/write -o bench -- $B --steps 2 --warmup 1 > $OUT/write_ov.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ov/trace -o bench -- $B --steps 5 --warmup 2 > $OUT/trace_ov.log 2>&1
find $OUT -name "*kernel_trace.csv" -size +20M -delete
du -sh $OUT
