#!/bin/bash
# A/B on one box: cap the scan kernel's blocks per CU through a dynamic LDS allocation (MOCAP_SCAN_LDS bytes)
for lds in 0 20480 32768 40960 53248 65536 0; do
  MOCAP_SCAN_LDS=$lds python bench.py --no-secondary --cpu-steps 0 > gpurun_out/lds_$lds.json 2>gpurun_out/lds_$lds.err || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/lds_$lds.json"))
print($lds, d["value"], d["ms_per_step"], d["roofline"].get("per_kernel"))
PY
done
