run() { lib=$1; args=$2; shift; shift
  cp $lib mocapv2_amd/libmocap_hip.so
  out=$(env "$@" python bench.py --no-secondary --no-extra --cpu-steps 0 --steps 20 $args 2>/dev/null | tail -1)
  echo "$(basename $lib) [$args] $* :: $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'alone', d['kernel_ms_per_step']['filter'], 'timed', d['kernel_ms_per_step_in_timed_region'])")"
}
cp mocapv2_amd/libmocap_hip.so /tmp/keep.so
B=scratch/libs/base.so; W=scratch/libs/rows_w4.so
for i in 1 2; do
  run $B "" MOCAP_WIDE_BLOCKS_PER_CU=3
  run $B "" MOCAP_WIDE_BLOCKS_PER_CU=2
  run $B "" MOCAP_WIDE_BLOCKS_PER_CU=1
  run $B "" MOCAP_WIDE_BLOCKS_PER_CU=2 MOCAP_WIDE_QUADS=40,40
  run $W "" MOCAP_WIDE_BLOCKS_PER_CU=3
  run $W "" MOCAP_WIDE_BLOCKS_PER_CU=2
  run $B "" MOCAP_ROWS_STAGED=1 MOCAP_WIDE_BLOCKS_PER_CU=2
  run $B "" MOCAP_ROWS_STAGED=1 MOCAP_WIDE_BLOCKS_PER_CU=1
done
for i in 1 2; do
  run $B "--markers 32" MOCAP_WIDE_BLOCKS_PER_CU=3
  run $B "--markers 32" MOCAP_WIDE_BLOCKS_PER_CU=2
  run $W "--markers 32" MOCAP_WIDE_BLOCKS_PER_CU=3
done
cp /tmp/keep.so mocapv2_amd/libmocap_hip.so
