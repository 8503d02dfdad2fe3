#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r02w2
mkdir -p $OUT
cd /tmp
export WHICH=wgrad SHAPES=0 REPS=3 VARIANT=${1:-2}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pmc1 -o p -- python $GRAFT_REPO_ROOT/scratch/bench_kernels.py > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum -d $OUT/pmc2 -o p -- python $GRAFT_REPO_ROOT/scratch/bench_kernels.py > $OUT/pmc2.log 2>&1
cd $GRAFT_REPO_ROOT
python scratch/pmc_report.py $OUT/pmc1 $OUT/pmc2 2>&1 | grep -v "^==" | tee gpurun_out/r02_pmc_wgrad_v$VARIANT.log
rm -f $OUT/*/*.db
