#!/bin/bash
# socket power and clocks (rocm-smi) sampled while the DeiT-S bs128 search step runs: is the chip at its power cap?
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -2
python bench.py --no-cpu-baseline --no-prof --steps 600 --warmup 5 > /tmp/bench_power.json 2>/dev/null &
BP=$!
for i in $(seq 1 40); do
  echo -n "t=$((i/2)).$((i%2*5))s "
  rocm-smi --showpower --showclocks 2>/dev/null | grep -i "Package Power\|sclk" | sed 's/.*: //' | tr '\n' ' '
  echo
  sleep 0.35
done
wait $BP
python -c "import json; d=json.load(open('/tmp/bench_power.json')); print('bench', d['ms_per_step'], 'ms/step')"
