#!/bin/bash
# power / clock samples while the training step runs (is the step power-limited?)
cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --agreement-n 0 --no-kernel-timer --steps 1500 --warmup 10 > /tmp/bench_long.json 2>/dev/null &
BP=$!
sleep 18
for i in 1 2 3 4 5 6 7 8 9 10; do
  rocm-smi --showpower --showclocks 2>&1 | grep -i "Power (W)\|sclk\|mclk" | sed 's/GPU\[0\]\s*: //' | tr '\n' ';'; echo
  sleep 0.5
done
wait $BP
tail -1 /tmp/bench_long.json | cut -c1-150
