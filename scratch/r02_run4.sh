#!/bin/bash
cd $GRAFT_REPO_ROOT
WHICH=fwd SHAPES=0 bash scratch/prof_wgrad.sh r02a > /dev/null 2>&1
python scratch/pmc_report.py gpurun_out/prof_r02a/pmc1 gpurun_out/prof_r02a/pmc2 gpurun_out/prof_r02a/pmc3 gpurun_out/prof_r02a/pmc4 gpurun_out/prof_r02a/trace 2>&1 | tee gpurun_out/r02_pmc4.log
export NBDT_HIP_LIB=$PWD/scratch/variants/libnbdt_pp_noepi.so
WHICH=fwd SHAPES=0 bash scratch/prof_wgrad.sh r02a_noepi > /dev/null 2>&1
python scratch/pmc_report.py gpurun_out/prof_r02a_noepi/pmc1 gpurun_out/prof_r02a_noepi/pmc3 2>&1 | tee gpurun_out/r02_pmc4_noepi.log
rm -rf gpurun_out/prof_r02a*/*/*/*.db 2>/dev/null
