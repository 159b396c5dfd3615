run() { args=$1; shift
  out=$(env "$@" python bench.py --no-secondary --no-extra --cpu-steps 0 --steps 12 $args 2>/dev/null | tail -1)
  echo "[$args] $* :: $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'alone', d['kernel_ms_per_step'], 'b2c', d['roofline']['blob_to_centroid']['frac'])")"
}
for i in 1 2; do
  run "--markers 32 --depth 1" MOCAP_WIDE_FORK=0
  run "--markers 32 --depth 1" MOCAP_WIDE_FORK=1
  run "--markers 32 --depth 1" MOCAP_WIDE_FORK=1 MOCAP_WIDE_BLOCKS_PER_CU=2
  run "--markers 32 --depth 1" MOCAP_WIDE_FORK=1 MOCAP_WIDE_BLOCKS_PER_CU=4
done
run "--markers 32" MOCAP_WIDE_FORK=0
run "--markers 32" MOCAP_WIDE_FORK=1
run "--markers 32" MOCAP_WIDE_FORK=0
run "--markers 32" MOCAP_WIDE_FORK=1
