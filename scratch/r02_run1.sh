#!/bin/bash
# round 2, GPU call 1: parity of the new ping-pong kernel, then kernel timings + ablation variants
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_backbone_gpu.py -x -q -m gpu -k "pingpong or bench_shape or conv_forward or epilogue" > gpurun_out/r02_t1.log 2>&1
echo "pytest conv rc=$?" >> gpurun_out/r02_t1.log
tail -5 gpurun_out/r02_t1.log
WHICH=fwd,dgrad,wgrad SHAPES=0,1,2 REPS=10 timeout 300 python scratch/bench_kernels.py > gpurun_out/r02_k1.log 2>&1
cat gpurun_out/r02_k1.log
SHAPES=0 WHICH=fwd timeout 300 bash scratch/ablate.sh pp_noepi pp_nodma pp_mfma pp_nomfma > gpurun_out/r02_abl1.log 2>&1
cat gpurun_out/r02_abl1.log
timeout 600 python -m pytest tests -x -q -m gpu > gpurun_out/r02_t2.log 2>&1
echo "pytest all rc=$?" >> gpurun_out/r02_t2.log
tail -15 gpurun_out/r02_t2.log
timeout 300 python bench.py --steps 10 --warmup 3 > gpurun_out/r02_b1.log 2>&1
tail -3 gpurun_out/r02_b1.log
