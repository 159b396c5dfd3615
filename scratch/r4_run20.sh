run() { args=$1; shift
  out=$(env "$@" python bench.py --no-secondary --no-extra --cpu-steps 0 --steps 20 $args 2>/dev/null | tail -1)
  echo "[$args] $* :: $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'timed', d['kernel_ms_per_step_in_timed_region'])")"
}
for i in 1 2; do
  run "" A=1
  run "" MOCAP_CONTOUR_BLOCKS_PER_CU=2 MOCAP_WIDE_BLOCKS_PER_CU=2
  run "" MOCAP_CONTOUR_BLOCKS_PER_CU=1 MOCAP_WIDE_BLOCKS_PER_CU=2
  run "" MOCAP_CONTOUR_BLOCKS_PER_CU=2 MOCAP_WIDE_BLOCKS_PER_CU=2 MOCAP_CORR_THREADS=128
  run "" MOCAP_SCAN_PRIO=3
  run "" MOCAP_SCAN_PRIO=3 MOCAP_CONTOUR_BLOCKS_PER_CU=2 MOCAP_WIDE_BLOCKS_PER_CU=2
done
