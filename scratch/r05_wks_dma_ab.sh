#!/bin/bash
# round 5: K-split weight-gradient kernel, LDS-DMA pieces issued in the load segment (shipped) vs at the head of the MFMA segment
cd /root/repo; O=gpurun_out/${1:-r05s}; mkdir -p $O
export NBDT_ALLOW_TIMING_BUILD=1
( NBDT_HIP_LIB=/root/repo/scratch/variants/libnbdt_wksdmam.so timeout 200 python -m pytest tests/test_backbone_gpu.py -k "weight_gradient_kernel_variants" -x -q 2>&1 | tail -2 ) > $O/pytest.txt 2>&1
for i in 1 2; do
  echo "# pieces in the load segment (this tree)"; WHICH=wgrad SHAPES=0,1,2 REPS=10 timeout 120 python scratch/bench_kernels.py
  echo "# pieces at the head of the MFMA segment (-DNBDT_WKS_DMA_IN_M)"; NBDT_HIP_LIB=/root/repo/scratch/variants/libnbdt_wksdmam.so WHICH=wgrad SHAPES=0,1,2 REPS=10 timeout 120 python scratch/bench_kernels.py
done > $O/wks_dma_ab.txt 2>&1
cat $O/pytest.txt $O/wks_dma_ab.txt
