#!/bin/bash
# A/B of environment settings on one box, two rounds: scratch/ab_env.sh "BENCH ARGS" "A=1" "B=2 C=3" ...
args=$1; shift
for i in 1 2; do
for cfg in "$@"; do
  out=$(env $cfg python bench.py --no-secondary --no-extra --cpu-steps 0 --steps 20 $args 2>/dev/null | tail -1)
  echo "$cfg :: $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'alone', d['kernel_ms_per_step'], 'timed', d['kernel_ms_per_step_in_timed_region'])")"
done
done
