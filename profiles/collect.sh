#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's roofline on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect.sh        -> gpurun_out/prof_{d3,d1,m32d1,fetch,write}/ ; then python profiles/summarize.py
# Counters go in their own passes, with --kernel-trace only (MI355X guide, HBM section).  The program follows `--` directly.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
set -e
ARGS="--cpu-steps 0 --no-secondary --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_d3 -o d3 -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_d3.json
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_d1 -o d1 -- python3 $R/bench.py --depth 1 $ARGS > $R/gpurun_out/prof_d1.json
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_m32d1 -o m32d1 -- python3 $R/bench.py --depth 1 --markers 32 $ARGS > $R/gpurun_out/prof_m32d1.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_fetch -o f -- python3 $R/bench.py --depth 1 --steps 3 --warmup 2 $ARGS > $R/gpurun_out/prof_fetch.json
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_write -o w -- python3 $R/bench.py --depth 1 --steps 3 --warmup 2 $ARGS > $R/gpurun_out/prof_write.json
# the same two passes for BASELINE.json configs[2] (32 markers)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_fetch32 -o f -- python3 $R/bench.py --depth 1 --markers 32 --steps 3 --warmup 2 $ARGS > $R/gpurun_out/prof_fetch32.json
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_write32 -o w -- python3 $R/bench.py --depth 1 --markers 32 --steps 3 --warmup 2 $ARGS > $R/gpurun_out/prof_write32.json
find $R/gpurun_out/prof_d3 $R/gpurun_out/prof_d1 $R/gpurun_out/prof_m32d1 $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write -name "*.csv" | head -30
