#!/bin/bash
# A/B of several builds of the library on one box: scratch/ab_libs.sh "BENCH ARGS" lib_a.so lib_b.so ...  (two rounds each)
cp mocapv2_amd/libmocap_hip.so /tmp/keep.so
args=$1; shift
for i in 1 2; do
  for lib in "$@"; do
    cp $lib mocapv2_amd/libmocap_hip.so
    python bench.py --cpu-steps 0 --no-secondary --no-extra --steps 20 $args 2> gpurun_out/t.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], 'alone', d['kernel_ms_per_step'], 'timed', d['kernel_ms_per_step_in_timed_region'])"
  done
done
cp /tmp/keep.so mocapv2_amd/libmocap_hip.so
