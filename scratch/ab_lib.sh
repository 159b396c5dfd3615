#!/bin/bash
# A/B of two builds of the library on one box: scratch/libmocap_hip_old.so against the current one
cp mocapv2_amd/libmocap_hip.so /tmp/new.so
for i in 1 2 3; do
  for v in old new; do
    if [ $v = old ]; then cp scratch/libmocap_hip_old.so mocapv2_amd/libmocap_hip.so; else cp /tmp/new.so mocapv2_amd/libmocap_hip.so; fi
    python bench.py --cpu-steps 0 --no-secondary 2> gpurun_out/t.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['kernel_ms_per_step']['contours'])"
  done
done
cp /tmp/new.so mocapv2_amd/libmocap_hip.so
