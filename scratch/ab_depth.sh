#!/bin/bash
# depth sweep with an environment setting: scratch/ab_depth.sh "ENV=.." 2 3 4 5
cfg=$1; shift
for d in "$@"; do
  out=$(env $cfg python bench.py --no-secondary --no-extra --cpu-steps 0 --steps 24 --depth $d 2>/dev/null | tail -1)
  echo "$cfg depth $d :: $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
done
