#!/bin/bash
# A/B of two builds of the library on one box: scratch/libmocap_hip_old.so against the current one (bench env in $1)
cp mocapv2_amd/libmocap_hip.so /tmp/new.so
for i in 1 2; do
  for v in old new; do
    if [ $v = old ]; then cp scratch/libmocap_hip_old.so mocapv2_amd/libmocap_hip.so; else cp /tmp/new.so mocapv2_amd/libmocap_hip.so; fi
    env $1 python bench.py --cpu-steps 0 --no-secondary --no-extra --steps 20 2> gpurun_out/t.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
  done
done
cp /tmp/new.so mocapv2_amd/libmocap_hip.so
