run() { # bench args, env...
  args=$1; shift
  out=$(env "$@" python bench.py --no-secondary --no-extra --cpu-steps 0 --steps 20 $args 2>/dev/null | tail -1)
  echo "[$args] $* :: $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'timed', d['kernel_ms_per_step_in_timed_region'])")"
}
for i in 1 2; do
  run "" A=1
  run "" GPU_MAX_HW_QUEUES=8
  run "--depth 4" GPU_MAX_HW_QUEUES=8
  run "--depth 5" GPU_MAX_HW_QUEUES=8
  run "--depth 4" A=1
  run "--depth 6" GPU_MAX_HW_QUEUES=16
done
python -m pytest tests/test_gpu_two_ranks.py -x -q 2>&1 | tail -5
python scratch/ab_camera_streams.py --markers 8 2>&1 | tail -1
python scratch/ab_camera_streams.py --markers 32 2>&1 | tail -1
