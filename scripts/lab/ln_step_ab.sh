#!/bin/bash
# Lab: the whole bs-128 step, previous build (scripts/lab/libofb_prev.so, not tracked) against the tree's build, interleaved on one box.
# usage (GPU box): bash scripts/lab/ln_step_ab.sh > gpurun_out/r06_ln_step_ab.txt
set -e
for i in 1 2 3; do
  for which in prev new; do
    if [ $which = prev ]; then export OFB_LIB_PATH=$PWD/scripts/lab/libofb_prev.so; else unset OFB_LIB_PATH; fi
    python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$which', d['value'], d['ms_per_step'])"
  done
done
