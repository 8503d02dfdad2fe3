#!/bin/bash
cd $GRAFT_REPO_ROOT
( echo "# s_memtime stamps around the segments of conv3x3_pp_kernel (build -DNBDT_PP_TIMING=1 -DNBDT_EPI_TIMING=1, scratch/pp_timing.py):"
  echo "# shader cycles per K step and wave, averaged over the launch; every column includes one stamp (~45 cycles);"
  echo "# an MFMA segment is 20 MFMAs = 640 cycles; epilogue phases in cycles per tile"
  NBDT_HIP_LIB=$PWD/scratch/variants/libnbdt_tim.so timeout 300 python scratch/pp_timing.py 2>&1 | grep -v amdgpu.ids ) | tee gpurun_out/r02_pp_segments.txt
( echo "# same for conv_wgrad_pp_kernel (build -DNBDT_WPP_TIMING=1, scratch/wpp_timing.py): cycles per 64-pixel stage;"
  echo "# MFMA segments: 50 (group 0) / 40 (group 1) v_mfma_f32_16x16x32_bf16 = 850 / 680 cycles"
  NBDT_HIP_LIB=$PWD/scratch/variants/libnbdt_wtim.so timeout 300 python scratch/wpp_timing.py 2>&1 | grep -v amdgpu.ids ) | tee gpurun_out/r02_wgrad_segments.txt
( echo "# scratch/bench_kernels.py: HIP-event time of 10 back-to-back launches per kernel, WRN-28-10 shapes at B=512, random bf16 operands"
  echo "# FORCE/VARIANT select kernels through the descriptors: igemm 2 = 8-wave ping-pong, 3 = 4-wave; wgrad 2 = 8-wave ping-pong, 3 = 4-wave"
  for F in 2 3; do echo "igemm wide_tile=$F wgrad variant=$F"; FORCE=$F VARIANT=$F WHICH=fwd,dgrad,wgrad SHAPES=0,1,2 REPS=10 timeout 300 python scratch/bench_kernels.py 2>&1 | grep -v amdgpu.ids; done
  echo "igemm ping-pong kernel without its epilogue (-DNBDT_PP_ABLATE=32):"
  SHAPES=0 WHICH=fwd FORCE=2 timeout 300 bash scratch/ablate.sh pp_noepi 2>&1 | grep -v amdgpu.ids ) | tee gpurun_out/r02_kernel_microbench.txt
