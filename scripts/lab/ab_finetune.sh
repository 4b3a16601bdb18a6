#!/bin/bash
# same-box A/B of library variants on the finetune bench (configs[4], ragged shapes): scripts/lab/ab_finetune.sh tagA tagB ...
cd "$(dirname "$0")/../.."
V=once-for-both_amd/csrc/build/variants
for rep in 1 2; do for tag in "$@"; do
  echo "$tag: $(OFB_LIB_PATH=$PWD/$V/libofb_hip.$tag.so python bench.py --mode finetune --batch 256 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])")"
done; done
