#!/bin/bash
# round 5: residual added in the epilogue's row walk (loads issued before the accumulator pass) vs through the transposition region
cd /root/repo; O=gpurun_out/${1:-r05p}; mkdir -p $O
(timeout 900 python -m pytest tests/test_backbone_gpu.py -x -q -k "not weight_gradient" 2>&1 | tail -6) > $O/pytest.txt 2>&1
for i in 1 2; do
  echo "# before (-DNBDT_PP_NO_PREFETCH: residual straight from HBM)"; NBDT_ALLOW_TIMING_BUILD=1 NBDT_HIP_LIB=/root/repo/scratch/variants/libnbdt_nopad.so WHICH=epi SHAPES=0,1,2 REPS=10 timeout 120 python scratch/bench_kernels.py
  echo "# this tree (residual prefetched through the caches during the K loop)";  WHICH=epi SHAPES=0,1,2 REPS=10 timeout 120 python scratch/bench_kernels.py
done > $O/res_ab.txt 2>&1
cat $O/pytest.txt $O/res_ab.txt
