#!/bin/bash
# box kernel at 20 / 14 / 11 KB of LDS per wave (8 / 11 / 14+ waves per CU): scratch/ab_box_lds.sh  (libs built by build_variant.sh)
run() { lib=$1; args=$2
  cp $lib mocapv2_amd/libmocap_hip.so
  out=$(python bench.py --no-secondary --no-extra --cpu-steps 0 --steps 20 $args 2>/dev/null | tail -1)
  echo "$(basename $lib) [$args] :: $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'alone filter', d['kernel_ms_per_step']['filter'], 'timed', d['kernel_ms_per_step_in_timed_region'])")"
}
cp mocapv2_amd/libmocap_hip.so /tmp/keep.so
for i in 1 2; do
  for l in base h1024 h768; do run scratch/libs/$l.so ""; done
done
for l in base h1024 h768; do run scratch/libs/$l.so "--markers 32"; done
cp /tmp/keep.so mocapv2_amd/libmocap_hip.so
