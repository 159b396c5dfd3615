run() { args=$1; shift
  out=$(env "$@" python bench.py --no-secondary --no-extra --cpu-steps 0 --steps 20 $args 2>/dev/null | tail -1)
  echo "[$args] $* :: $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'alone', d['kernel_ms_per_step']['contours'], 'timed', d['kernel_ms_per_step_in_timed_region'])")"
}
python -m pytest tests/test_gpu_blob.py -x -q 2>&1 | tail -2
for i in 1 2 3; do
  run "" MOCAP_CONTOUR_DEFER=2
  run "" MOCAP_CONTOUR_DEFER=1
done
run "--markers 32" MOCAP_CONTOUR_DEFER=2
run "--markers 32" MOCAP_CONTOUR_DEFER=0
run "--markers 32" MOCAP_CONTOUR_DEFER=2
run "--markers 32" MOCAP_CONTOUR_DEFER=0
