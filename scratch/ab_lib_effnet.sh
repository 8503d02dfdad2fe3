#!/bin/bash
# same-box A/B of the EfficientNet-B0 step (config 5) between libraries: ab_lib_effnet.sh <outfile> <rounds> lib1.so lib2.so ... (HEAD = this tree's)
cd /root/repo; OUT=$1; ROUNDS=$2; shift 2
mkdir -p $(dirname $OUT)
export NBDT_ALLOW_TIMING_BUILD=1
for i in $(seq $ROUNDS); do
  for L in "$@"; do
    if [ "$L" != "HEAD" ]; then T=$(basename $L); export NBDT_HIP_LIB=/root/repo/$L; else T="this tree"; unset NBDT_HIP_LIB; fi
    timeout 200 python scratch/bench_effnet.py --steps 30 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s %.1f img/s  %.3f ms/step' % ('$T', d['img_per_s'], d['ms_per_step']))"
  done
done > $OUT 2>&1
cat $OUT
