#!/bin/bash
# round 4's library + host code (scratch/variants/r04tree, built from f41217f) against HEAD on the
# three headline shapes (forward + statistics, forward + residual + statistics, plain data gradient) and the whole step.
# Same box, alternating, 3 rounds.
OUT=gpurun_out/r06_r04_vs_head_igemm_ab.txt
R04=scratch/variants/r04tree
{
echo "# $(date -u) $(rocm-smi --showproductname 2>/dev/null | grep -m1 'Card Series' )"
for R in 1 2 3; do
  echo "== round $R: r04 (f41217f)"
  (cd $R04 && WHICH=epi,dgrad SHAPES=0,1,2 REPS=10 python scratch/bench_kernels.py)
  echo "== round $R: HEAD"
  WHICH=epi,dgrad SHAPES=0,1,2 REPS=10 python scratch/bench_kernels.py
done
for R in 1 2 3; do
  echo -n "step r04   "; (cd $R04 && python bench.py --no-cpu-baseline --no-kernel-timer --agreement-n 0 --no-other-configs --steps 30 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  echo -n "step HEAD  "; python bench.py --no-cpu-baseline --no-kernel-timer --agreement-n 0 --no-other-configs --steps 30 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
} > $OUT 2>&1
tail -50 $OUT
