#!/bin/bash
# A/B of bench arguments on one box: scratch/ab_args.sh "--depth 3" "--depth 4" ...
for i in 1 2; do
for cfg in "$@"; do
  out=$(python bench.py --no-secondary --no-extra --cpu-steps 0 --steps 20 $cfg 2>/dev/null | tail -1)
  echo "$cfg :: $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'])")"
done
done
