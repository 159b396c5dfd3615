#!/bin/bash
# A/B of environment switches on one box: scratch/ab_bench.sh "VAR=val VAR2=val" ... ; prints value / ms_per_step / kernel times per setting
for cfg in "$@"; do
  out=$(env $cfg python bench.py --no-secondary --no-extra --cpu-steps 0 --steps 20 2>/dev/null | tail -1)
  echo "$cfg :: $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['dark_tile_early_out']['tiles_per_step']-d['dark_tile_early_out']['tiles_resolved_without_filtering'])")"
done
