#!/usr/bin/env python3
"""Turns the rocprofv3 output of profiles/collect.sh (gpurun_out/prof_*) into the summaries kept under profiles/:
rNN_kernel_stats_depth3_default.csv, rNN_kernel_stats_depth1(_markers32).csv (copies of rocprofv3's own kernel_stats),
rNN_pmc_FETCH_SIZE(_markers32).csv / rNN_pmc_WRITE_SIZE(_markers32).csv (per-kernel averages in KB per launch) and hbm_traffic.json
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
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


ROUND = "r04"
SHORT = ("bright_cells_kernel", "mark_tiles_kernel", "settle_tiles_kernel", "box_filter_kernel", "filter_rows_staged_kernel", "filter_mask_kernel")


def workload(tag_fetch, tag_write, suffix, key, markers):
    fdir, wdir = os.path.join(SRC, tag_fetch), os.path.join(SRC, tag_write)
    if not (os.path.isdir(fdir) and os.path.isdir(wdir)):
        return None
    fetch, counts_f = pmc(os.path.join(fdir, "f_counter_collection.csv"), "FETCH_SIZE", os.path.join(OUT, f"{ROUND}_pmc_FETCH_SIZE{suffix}.csv"))
    write, counts_w = pmc(os.path.join(wdir, "w_counter_collection.csv"), "WRITE_SIZE", os.path.join(OUT, f"{ROUND}_pmc_WRITE_SIZE{suffix}.csv"))
    per = {}
    for short in SHORT:
        kf = [k for k in fetch if short in k]
        kw = [k for k in write if short in k]
        if not kf and not kw:
            continue
        # (two instances of a template can appear in one run: the scan marks the tiles itself until the first probe has counted
        # the hot cells, then leaves the hot map -- both read the same bytes; the one launched most often is taken)
        kf.sort(key=lambda k: -counts_f[k]); kw.sort(key=lambda k: -counts_w[k])
        per[short] = int(round((2 * fetch[kf[0]] + write[kw[0]]) * 1024))
    scan_fetch = fetch[sorted((k for k in fetch if "bright_cells_kernel" in k), key=lambda k: -counts_f[k])[0]]
    algo = FRAME_BYTES * IMAGES
    total = sum(per.values())
    print(key, json.dumps(per), "total", total, "= %.3f x algorithmic" % (total / algo))
    return {"workload_key": key, "dist": "mild", "images_per_launch": float(IMAGES), "markers": markers, "hbm_bytes_per_launch": per,
            "total_over_algorithmic": round(total / algo, 4),
            "derivation": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 from profiles/%s_pmc_FETCH_SIZE%s.csv / %s_pmc_WRITE_SIZE%s.csv (separate --pmc "
                          "passes of `python bench.py --depth 1 %s--steps 3 --warmup 2 --cpu-steps 0 --no-secondary --no-extra`, "
                          "profiles/collect.sh + summarize.py); FETCH_SIZE doubled per the MI355X guide's gfx950 rule, confirmed on this "
                          "access pattern: bright_cells_kernel reads every frame byte exactly once (%d B per launch, 16-byte loads) and "
                          "FETCH_SIZE reports %.1f KB = %.4f of it" % (ROUND, suffix, ROUND, suffix, "--markers 32 " if markers == 32 else "",
                                                                     algo, scan_fetch, scan_fetch * 1024 / algo)}


def main():
    for tag, name in (("prof_d3", "depth3_default"), ("prof_d1", "depth1"), ("prof_m32d1", "depth1_markers32")):
        src = os.path.join(SRC, tag, tag[5:] + "_kernel_stats.csv")
        if os.path.exists(src):
            shutil.copy(src, os.path.join(OUT, f"{ROUND}_kernel_stats_{name}.csv"))
    docs = {}
    for args in (("prof_fetch", "prof_write", "", "6x1920x1080-m8-mild", 8), ("prof_fetch32", "prof_write32", "_markers32", "6x1920x1080-m32-mild", 32)):
        d = workload(*args)
        if d:
            docs[d["workload_key"]] = d
    with open(os.path.join(OUT, "hbm_traffic.json"), "w") as f:
        json.dump({"workloads": docs}, f, indent=1)


if __name__ == "__main__":
    main()
