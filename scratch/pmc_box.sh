#!/bin/bash
# PMC passes over a short depth-1 bench run: per-kernel averages of a few SQ counters (scratch/pmc_box.sh)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -o p -- python3 $R/bench.py --depth 1 --steps 2 --warmup 1 --cpu-steps 0 --no-secondary --no-extra > /dev/null 2> $R/gpurun_out/pmc_$tag.err || echo "pass failed: $grp"
  python3 - "$R/gpurun_out/pmc_$tag" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-40:]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    if any(s in k for s in ("box_filter", "bright_cells", "contours", "settle")):
        print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
done
