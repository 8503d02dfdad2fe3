#!/bin/bash
# round 5: weight-gradient kernel A/B on one box -- round-4 kernel (scratch/variants/libnbdt_wgrad_r4.so) vs the tree's
cd /root/repo; O=gpurun_out/${1:-r05b}; mkdir -p $O
(timeout 600 python -m pytest tests/test_backbone_gpu.py -k "weight_gradient or bench_shape or wgrad" -x -q 2>&1 | tail -8) > $O/pytest.txt 2>&1
for i in 1 2; do
  echo "# round-4 kernel"; NBDT_HIP_LIB=/root/repo/scratch/variants/libnbdt_wgrad_r4.so WHICH=wgrad SHAPES=0,1,2 REPS=10 timeout 120 python scratch/bench_kernels.py
  echo "# this tree";     WHICH=wgrad SHAPES=0,1,2 REPS=10 timeout 120 python scratch/bench_kernels.py
done > $O/wgrad_ab.txt 2>&1
if [ -f scratch/variants/libnbdt_wtim1.so ]; then
  NBDT_ALLOW_TIMING_BUILD=1 NBDT_HIP_LIB=/root/repo/scratch/variants/libnbdt_wtim1.so timeout 120 python scratch/wpp_timing.py > $O/wpp_timing.txt 2>&1
fi
cat $O/pytest.txt $O/wgrad_ab.txt $O/wpp_timing.txt
