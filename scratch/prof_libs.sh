#!/bin/bash
# per-kernel durations (rocprofv3 kernel trace, depth 1) of several builds on one box: scratch/prof_libs.sh "BENCH ARGS" lib_a.so lib_b.so ...
cp mocapv2_amd/libmocap_hip.so /tmp/keep.so
args=$1; shift
for lib in "$@"; do
  cp $lib mocapv2_amd/libmocap_hip.so
  tag=$(basename $lib .so)
  echo "== $lib $args"
  bash scratch/prof_one.sh $tag --depth 1 $args | grep -v "remap_\|undistort_map\|rocclr\|srcbox"
done
cp /tmp/keep.so mocapv2_amd/libmocap_hip.so
