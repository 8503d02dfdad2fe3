#!/bin/bash
# round 5: padded LDS pitch (conv3x3_pp_kernel<.., PAD>) -- parity tests, then same-box A/B against -DNBDT_PP_NO_PAD
cd /root/repo; O=gpurun_out/${1:-r05f}; mkdir -p $O
(timeout 900 python -m pytest tests/test_backbone_gpu.py -x -q -k "not weight_gradient" 2>&1 | tail -8) > $O/pytest.txt 2>&1
for i in 1 2; do
  echo "# contiguous halo image (-DNBDT_PP_NO_PAD)"; NBDT_ALLOW_TIMING_BUILD=1 NBDT_HIP_LIB=/root/repo/scratch/variants/libnbdt_nopad.so WHICH=fwd,dgrad,epi SHAPES=1,2 FORCE=2 REPS=10 timeout 120 python scratch/bench_kernels.py
  echo "# padded pitch (this tree)";                 WHICH=fwd,dgrad,epi SHAPES=1,2 FORCE=2 REPS=10 timeout 120 python scratch/bench_kernels.py
done > $O/pad_ab.txt 2>&1
cat $O/pytest.txt $O/pad_ab.txt
