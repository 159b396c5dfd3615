#!/bin/bash
# rocprofv3 kernel trace of one bench configuration: scratch/prof_one.sh <tag> <bench args...>  -> gpurun_out/prof_<tag>/<tag>_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o $tag -- python3 $R/bench.py --cpu-steps 0 --no-secondary --no-extra --steps 10 --warmup 3 "$@" > $R/gpurun_out/prof_$tag.json 2> $R/gpurun_out/prof_$tag.err
f=$(find $R/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-70s calls %6s avg_us %10.1f total_ms %9.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
