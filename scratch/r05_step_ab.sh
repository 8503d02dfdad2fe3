#!/bin/bash
# round 5: whole-step same-box A/B: library with round 4's weight-gradient kernel vs this tree (K-split kernel), alternating
cd /root/repo; O=gpurun_out/${1:-r05n}; mkdir -p $O
ARGS="--no-cpu-baseline --no-other-configs --agreement-n 0 --no-attainable --steps 40 --warmup 5"
for i in 1 2 3; do
  for L in /root/repo/scratch/variants/libnbdt_wgrad_r4.so ""; do
    if [ -n "$L" ]; then T="round-4 wgrad kernel"; export NBDT_HIP_LIB=$L; else T="this tree"; unset NBDT_HIP_LIB; fi
    timeout 200 python bench.py $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
w = d['roofline_wgrad']
print('$T: %.1f img/s  %.3f ms/step   wgrad alone %.1f us, in step %.1f us   igemm %.1f us' % (d['value'], d['ms_per_step'], w['avg_launch_us'], w['in_step_us'], d['roofline']['avg_launch_us']))"
  done
done > $O/step_ab.txt 2>&1
cat $O/step_ab.txt
