#!/bin/bash
# step time (ms) of the DeiT-S bs128 search step over {hardware queues} x {side-stream priority} x {eager, graph} x {plain, one-rank RCCL}
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
B="python bench.py --no-cpu-baseline --no-prof --steps 15 --warmup 4"
for q in 2 4 8; do for pr in 0 -1; do
  line="queues=$q prio=$pr:"
  for mode in "" "--force-dp" "--graph" "--graph --force-dp"; do
    v=$(GPU_MAX_HW_QUEUES=$q OFB_SIDE_PRIORITY=$pr $B $mode 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
    line="$line  [${mode:-eager}] ${v:-fail}"
  done
  echo "$line"
done; done
