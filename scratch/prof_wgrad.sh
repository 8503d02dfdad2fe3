#!/bin/bash
# rocprofv3 passes over the kernel micro-benchmark (run on the GPU box)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$1
mkdir -p $OUT
cd /tmp
export WHICH=${WHICH:-wgrad,fwd} SHAPES=${SHAPES:-0} REPS=3
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/scratch/bench_kernels.py > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $OUT/pmc1 -o p -- python $R/scratch/bench_kernels.py > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM -d $OUT/pmc2 -o p -- python $R/scratch/bench_kernels.py > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc3 -o p -- python $R/scratch/bench_kernels.py > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc4 -o p -- python $R/scratch/bench_kernels.py > $OUT/pmc4.log 2>&1
find $OUT -name "*.csv" | head -30
