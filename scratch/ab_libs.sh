#!/bin/bash
# A/B of several builds of the library on one box: scratch/ab_libs.sh "ENV=.." lib_a.so lib_b.so ...  (two rounds each)
cp mocapv2_amd/libmocap_hip.so /tmp/keep.so
envs=$1; shift
for i in 1 2; do
  for lib in "$@"; do
    cp $lib mocapv2_amd/libmocap_hip.so
    env $envs python bench.py --cpu-steps 0 --no-secondary --no-extra --steps 20 2> gpurun_out/t.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
  done
done
cp /tmp/keep.so mocapv2_amd/libmocap_hip.so
