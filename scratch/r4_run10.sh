run() { # bench args, env...
  args=$1; shift
  out=$(env "$@" python bench.py --no-secondary --no-extra --cpu-steps 0 --steps 20 $args 2>/dev/null | tail -1)
  echo "[$args] $* :: $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'timed', d['kernel_ms_per_step_in_timed_region'])")"
}
for i in 1 2; do
  run "" A=1
  for b in 3 4 6; do
    run "--depth 2" MOCAP_SCAN_BLOCKS_PER_CU=$b
    run "--depth 2" MOCAP_SCAN_BLOCKS_PER_CU=$b MOCAP_SCAN_HOTMAP=0
  done
  run "--depth 2" MOCAP_SCAN_BLOCKS_PER_CU=4 MOCAP_WIDE_BLOCKS_PER_CU=2
  run "--depth 3" MOCAP_SCAN_BLOCKS_PER_CU=4 MOCAP_SCAN_HOTMAP=0
done
