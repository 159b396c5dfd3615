#!/usr/bin/env python3
"""Steady-state timeline of the pipelined bench from a rocprofv3 kernel trace: which kernel ran when, on which hardware queue,
for how long.  python profiles/timeline.py gpurun_out/prof_d3/d3_kernel_trace.csv [first scan to start at] [kernels to print]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = ("remap", "undistort", "srcbox", "rowbox")
ks = [r for r in rows if "mocap" in r["Kernel_Name"] and not any(s in r["Kernel_Name"] for s in skip)]
ks.sort(key=lambda r: int(r["Start_Timestamp"]))
scans = [i for i, r in enumerate(ks) if "bright_cells" in r["Kernel_Name"]]
first = scans[int(sys.argv[2]) if len(sys.argv) > 2 else len(scans) // 3]
t0 = int(ks[first]["Start_Timestamp"])
print("queue kernel                            start us ->   end us   duration us   workgroups")
for r in ks[first:first + (int(sys.argv[3]) if len(sys.argv) > 3 else 40)]:
    n = r["Kernel_Name"].split("(")[0].replace("void mocap::", "").replace("mocap::", "")[:32]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    wg = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    print(f"q{r['Queue_Id']}    {n:32s} {s:9.1f} -> {e:9.1f}   {e - s:9.1f}   {wg}")
