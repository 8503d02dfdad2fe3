#!/bin/bash
# same-box A/B of whole-step time between libraries: ab_lib_step.sh <outfile> <rounds> lib1.so lib2.so ...   ("" = this tree's library)
cd /root/repo; OUT=$1; ROUNDS=$2; shift 2
mkdir -p $(dirname $OUT)
ARGS="--no-cpu-baseline --no-other-configs --agreement-n 0 --no-attainable --no-kernel-timer --steps 40 --warmup 5"
export NBDT_ALLOW_TIMING_BUILD=1
for i in $(seq $ROUNDS); do
  for L in "$@"; do
    if [ "$L" != "HEAD" ]; then T=$(basename $L); export NBDT_HIP_LIB=/root/repo/$L; else T="this tree"; unset NBDT_HIP_LIB; fi
    timeout 200 python bench.py $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s %.1f img/s  %.3f ms/step' % ('$T', d['value'], d['ms_per_step']))"
  done
done > $OUT 2>&1
cat $OUT
