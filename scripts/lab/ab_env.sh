#!/bin/bash
# same-box A/B of environment settings on the default search bench: scripts/lab/ab_env.sh "A=1" "A=2 B=3" ...
cd "$(dirname "$0")/../.."
for rep in 1 2 3; do for e in "$@"; do
  echo "$e: $(env $e python bench.py --no-cpu-baseline --no-prof --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")"
done; done
