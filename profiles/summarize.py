#!/usr/bin/env python3
"""Turns the rocprofv3 output of profiles/collect.sh (gpurun_out/prof_*) into the summaries kept under profiles/:
r03_kernel_stats_depth3_default.csv, r03_kernel_stats_depth1.csv (copies of rocprofv3's own kernel_stats),
r03_pmc_FETCH_SIZE.csv / r03_pmc_WRITE_SIZE.csv (per-kernel averages in KB per launch) and hbm_traffic.json
(= (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch; FETCH_SIZE doubled per the MI355X guide's gfx950 rule)."""
import csv
import json
import os
import shutil
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "profiles")
SRC = os.path.join(ROOT, "gpurun_out")
FRAME_BYTES, IMAGES = 1920 * 1080, 3072


def pmc(path, counter, dst):
    acc = OrderedDict()
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            acc.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    with open(dst, "w") as f:
        f.write("kernel,launches,avg_KB,min_KB,max_KB\n")
        for k, v in acc.items():
            f.write('"%s",%d,%.3f,%.3f,%.3f\n' % (k, len(v), sum(v) / len(v), min(v), max(v)))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    shutil.copy(os.path.join(SRC, "prof_d3", "d3_kernel_stats.csv"), os.path.join(OUT, "r03_kernel_stats_depth3_default.csv"))
    shutil.copy(os.path.join(SRC, "prof_d1", "d1_kernel_stats.csv"), os.path.join(OUT, "r03_kernel_stats_depth1.csv"))
    if os.path.exists(os.path.join(SRC, "prof_m32d1", "m32d1_kernel_stats.csv")):  # BASELINE.json configs[2]: 32 markers
        shutil.copy(os.path.join(SRC, "prof_m32d1", "m32d1_kernel_stats.csv"), os.path.join(OUT, "r03_kernel_stats_depth1_markers32.csv"))
    fetch = pmc(os.path.join(SRC, "prof_fetch", "f_counter_collection.csv"), "FETCH_SIZE", os.path.join(OUT, "r03_pmc_FETCH_SIZE.csv"))
    write = pmc(os.path.join(SRC, "prof_write", "w_counter_collection.csv"), "WRITE_SIZE", os.path.join(OUT, "r03_pmc_WRITE_SIZE.csv"))
    per = {}
    for short in ("bright_cells_kernel", "settle_tiles_kernel", "box_filter_kernel", "filter_mask_kernel"):
        kf = [k for k in fetch if short in k]
        kw = [k for k in write if short in k]
        if not kf and not kw:
            continue
        assert len(kf) == 1 and len(kw) == 1, (short, kf, kw)
        per[short] = int(round((2 * fetch[kf[0]] + write[kw[0]]) * 1024))
    scan_fetch = fetch[[k for k in fetch if "bright_cells_kernel" in k][0]]
    algo = FRAME_BYTES * IMAGES
    doc = {"workload_key": "6x1920x1080-m8-mild", "dist": "mild", "images_per_launch": float(IMAGES), "markers": 8, "hbm_bytes_per_launch": per,
           "derivation": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 from profiles/r03_pmc_FETCH_SIZE.csv / r03_pmc_WRITE_SIZE.csv (separate "
                         "--pmc passes of `python bench.py --depth 1 --steps 3 --warmup 2 --cpu-steps 0 --no-secondary --no-extra`, "
                         "profiles/collect.sh + summarize.py); FETCH_SIZE doubled per the MI355X guide's gfx950 rule, confirmed on this "
                         "access pattern: bright_cells_kernel reads every frame byte exactly once (%d B per launch, 16-byte loads) "
                         "and FETCH_SIZE reports %.1f KB = %.4f of it" % (algo, scan_fetch, scan_fetch * 1024 / algo)}
    with open(os.path.join(OUT, "hbm_traffic.json"), "w") as f:
        json.dump(doc, f, indent=1)
    total = sum(per.values())
    print(json.dumps(per), "total", total, "= %.3f x algorithmic" % (total / algo))


if __name__ == "__main__":
    main()
