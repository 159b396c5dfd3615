#!/bin/bash
# the follow kernel's phase clock for several builds on one box: scratch/follow_clock.sh "BENCH ARGS" lib_a.so ...
cp mocapv2_amd/libmocap_hip.so /tmp/keep.so
args=$1; shift
for lib in "$@"; do
  cp $lib mocapv2_amd/libmocap_hip.so
  echo "== $lib $args"
  MOCAP_FOLLOW_TIMING=1 timeout -k 10 120 python bench.py --depth 1 --steps 2 --warmup 1 --cpu-steps 0 --no-secondary --no-extra $args 2>&1 >/dev/null | grep follow | tail -1
  timeout -k 10 120 python bench.py --depth 1 --steps 10 --warmup 3 --cpu-steps 0 --no-secondary --no-extra $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   contours alone ms', d['kernel_ms_per_step']['contours'])"
done
cp /tmp/keep.so mocapv2_amd/libmocap_hip.so
