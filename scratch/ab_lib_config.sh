#!/bin/bash
# same-box A/B of one BASELINE configuration between libraries: ab_lib_config.sh <config c1|c3|c4|c5> <outfile> <rounds> lib1.so ... (HEAD = this tree's)
cd /root/repo; C=$1; OUT=$2; ROUNDS=$3; shift 3
mkdir -p $(dirname $OUT)
export NBDT_ALLOW_TIMING_BUILD=1
for i in $(seq $ROUNDS); do
  for L in "$@"; do
    if [ "$L" != "HEAD" ]; then T=$(basename $L); export NBDT_HIP_LIB=/root/repo/$L; else T="this tree"; unset NBDT_HIP_LIB; fi
    timeout 200 python scratch/run_config.py $C --steps 40 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-4s %-28s %.1f img/s  %.3f ms/step' % ('$C', '$T', d['img_per_s'], d['ms']))"
  done
done > $OUT 2>&1
cat $OUT
