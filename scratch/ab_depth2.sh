#!/bin/bash
# depth x environment grid on one box: scratch/ab_depth2.sh "ENV SETTINGS" depth...   (two rounds)
cfg=$1; shift
for i in 1 2; do
for d in "$@"; do
  out=$(env $cfg python bench.py --no-secondary --no-extra --cpu-steps 0 --steps 24 --depth $d 2>/dev/null | tail -1)
  echo "depth $d $cfg :: $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'timed', d['kernel_ms_per_step_in_timed_region'])")"
done
done
