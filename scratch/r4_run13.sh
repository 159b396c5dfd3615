run() { # bench args, env...
  args=$1; shift
  out=$(env "$@" python bench.py --no-secondary --no-extra --cpu-steps 0 --steps 20 $args 2>/dev/null | tail -1)
  echo "[$args] $* :: $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'alone', d['kernel_ms_per_step']['contours'], 'timed', d['kernel_ms_per_step_in_timed_region'])")"
}
python -m pytest tests/test_gpu_blob.py -x -q -k "contours or tuning or borders" 2>&1 | tail -2
MOCAP_CONTOUR_BLOCKS_PER_CU=2 python -m pytest tests/test_gpu_blob.py tests/test_gpu_oracle_e2e.py -x -q -k "contours or borders or configs2 or replay" 2>&1 | tail -2
for i in 1 2; do
  run "" MOCAP_CONTOUR_BLOCKS_PER_CU=0
  run "" MOCAP_CONTOUR_BLOCKS_PER_CU=2
  run "" MOCAP_CONTOUR_BLOCKS_PER_CU=3
  run "" MOCAP_CONTOUR_BLOCKS_PER_CU=4
  run "" MOCAP_CONTOUR_BLOCKS_PER_CU=6
done
run "--markers 32" MOCAP_CONTOUR_BLOCKS_PER_CU=0
run "--markers 32" MOCAP_CONTOUR_BLOCKS_PER_CU=3
run "--markers 32" MOCAP_CONTOUR_BLOCKS_PER_CU=0
run "--markers 32" MOCAP_CONTOUR_BLOCKS_PER_CU=3
