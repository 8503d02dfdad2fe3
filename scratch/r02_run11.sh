#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r02w
mkdir -p $OUT
cd /tmp
export WHICH=wgrad SHAPES=0 REPS=3
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/pmc1 -o p -- python $GRAFT_REPO_ROOT/scratch/bench_kernels.py > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM -d $OUT/pmc2 -o p -- python $GRAFT_REPO_ROOT/scratch/bench_kernels.py > $OUT/pmc2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum -d $OUT/pmc3 -o p -- python $GRAFT_REPO_ROOT/scratch/bench_kernels.py > $OUT/pmc3.log 2>&1
rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_TAGCONFLICT_STALL_CYCLES_sum -d $OUT/pmc4 -o p -- python $GRAFT_REPO_ROOT/scratch/bench_kernels.py > $OUT/pmc4.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $OUT/pmc5 -o p -- python $GRAFT_REPO_ROOT/scratch/bench_kernels.py > $OUT/pmc5.log 2>&1
cd $GRAFT_REPO_ROOT
python scratch/pmc_report.py $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 $OUT/pmc5 2>&1 | grep -v "^==" | tee gpurun_out/r02_pmc_wgrad.log
tail -3 $OUT/pmc3.log $OUT/pmc4.log | grep -i "error\|invalid\|not" | head
rm -f $OUT/*/*.db
