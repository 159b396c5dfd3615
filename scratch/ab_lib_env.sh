#!/bin/bash
# library x environment grid on one box, two rounds: scratch/ab_lib_env.sh "lib_a.so lib_b.so" "BENCH ARGS" "ENV1" "ENV2" ...
cp mocapv2_amd/libmocap_hip.so /tmp/keep.so
libs=$1; shift
args=$1; shift
for i in 1 2; do
  for lib in $libs; do
    cp $lib mocapv2_amd/libmocap_hip.so
    for cfg in "$@"; do
      out=$(env $cfg python bench.py --no-secondary --no-extra --cpu-steps 0 --steps 20 $args 2>/dev/null | tail -1)
      echo "$lib $cfg :: $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'alone', d['kernel_ms_per_step'], 'timed', d['kernel_ms_per_step_in_timed_region'])")"
    done
  done
done
cp /tmp/keep.so mocapv2_amd/libmocap_hip.so
