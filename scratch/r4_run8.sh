run() { # lib, bench args, env...
  lib=$1; args=$2; shift; shift
  cp $lib mocapv2_amd/libmocap_hip.so
  out=$(env "$@" python bench.py --no-secondary --no-extra --cpu-steps 0 --steps 20 $args 2>/dev/null | tail -1)
  echo "$lib [$args] $* :: $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'timed', d['kernel_ms_per_step_in_timed_region'])")"
}
cp mocapv2_amd/libmocap_hip.so /tmp/keep.so
L=scratch/libs/lean.so; B=scratch/libs/base.so
for i in 1 2; do
  run $L "" MOCAP_SCAN_SERIAL=1
  run $B "" MOCAP_SCAN_SERIAL=1
  run $L "" MOCAP_SCAN_SERIAL=1 MOCAP_SCAN_HOTMAP=0
  run $L "" MOCAP_SCAN_SERIAL=1 MOCAP_CONTOUR_PRIO=3
  run $L "" MOCAP_SCAN_SERIAL=1 MOCAP_CONTOUR_PRIO=3 MOCAP_BOX_PRIO=1 MOCAP_CORR_PRIO=3
  run $L "--depth 4" MOCAP_SCAN_SERIAL=1
  run $L "--depth 4" MOCAP_SCAN_SERIAL=1 MOCAP_CONTOUR_PRIO=3
  run $L "--depth 2" MOCAP_SCAN_SERIAL=1
  run $L "" MOCAP_SCAN_SERIAL=1 MOCAP_WIDE_BLOCKS_PER_CU=3 MOCAP_WIDE_QUADS=34,34
  run $L "" MOCAP_SCAN_SERIAL=1 MOCAP_WIDE_BLOCKS_PER_CU=3
done
echo "--- 32 markers"
for i in 1 2; do
  run $L "--markers 32" MOCAP_SCAN_SERIAL=0
  run $L "--markers 32" MOCAP_SCAN_SERIAL=1
  run $L "--markers 32" MOCAP_SCAN_SERIAL=1 MOCAP_CONTOUR_PRIO=3
  run $L "--markers 32" MOCAP_SCAN_SERIAL=1 MOCAP_WIDE_BLOCKS_PER_CU=3 MOCAP_WIDE_QUADS=34,34
  run $L "--markers 32 --depth 4" MOCAP_SCAN_SERIAL=1
done
cp /tmp/keep.so mocapv2_amd/libmocap_hip.so
