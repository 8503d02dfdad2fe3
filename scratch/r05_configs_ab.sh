#!/bin/bash
# round 5: other configurations with different weight-gradient kernel selections, same box
cd /root/repo; O=gpurun_out/${1:-r05r}; mkdir -p $O
(timeout 600 python -m pytest tests/test_backbone_gpu.py tests/test_models_gpu.py -x -q -k "weight_gradient or resnet or ResNet" 2>&1 | tail -4) > $O/pytest.txt 2>&1
for i in 1 2; do
 for L in wgrad_r4 "" wmin128 wmin64; do
  if [ -n "$L" ]; then export NBDT_ALLOW_TIMING_BUILD=1 NBDT_HIP_LIB=/root/repo/scratch/variants/libnbdt_$L.so; T=$L; else unset NBDT_HIP_LIB; T=tree; fi
  for c in c1 c3 c4; do echo -n "$T $c: "; timeout 120 python scratch/run_config.py $c --steps 30 --warmup 5 2>/dev/null | tail -1; done
 done
done > $O/configs_ab.txt 2>&1
cat $O/pytest.txt $O/configs_ab.txt
