#!/bin/bash
# round 5: 12-wave (three-group) weight-gradient kernel: parity tests, then VARIANT=2 (8-wave) vs VARIANT=4 on one box
cd /root/repo; O=gpurun_out/${1:-r05k}; mkdir -p $O
(timeout 300 python -m pytest tests/test_backbone_gpu.py -k "weight_gradient_kernel_variants" -x -q 2>&1 | tail -8) > $O/pytest.txt 2>&1
for i in 1 2; do
  for v in 2 5; do echo "# VARIANT=$v"; VARIANT=$v WHICH=wgrad SHAPES=0,1,2 REPS=10 timeout 120 python scratch/bench_kernels.py; done
done > $O/wgrad3_ab.txt 2>&1
cat $O/pytest.txt $O/wgrad3_ab.txt
