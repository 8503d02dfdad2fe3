#!/bin/bash
# same-box A/B of the whole training step: env settings given as arguments, alternating, 2 rounds
for R in 1 2; do
  for E in "$@"; do
    echo -n "$E  "; env $E python bench.py --no-cpu-baseline --no-kernel-timer --agreement-n 0 --no-other-configs --steps 20 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done
