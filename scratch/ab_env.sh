#!/bin/bash
# A/B of environment settings on one box, two rounds: scratch/ab_env.sh "A=1" "B=2 C=3" ...
for i in 1 2; do
for cfg in "$@"; do
  out=$(env $cfg python bench.py --no-secondary --no-extra --cpu-steps 0 --steps 20 2>/dev/null | tail -1)
  echo "$cfg :: $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'])")"
done
done
