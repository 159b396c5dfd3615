"""Timing probe: filter-stage kernels on all-dark frames vs the bench frames (not a test, not shipped)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mocapv2_amd.pipeline import BatchTracker, scene_arrays
from mocapv2_amd.synth import MILD_DIST, ZERO_DIST, Scene

W, H, C, T = 1920, 1080, 6, 64
for name, dist in (("mild", MILD_DIST), ("zero", ZERO_DIST)):
    sc = Scene(C, W, H, dist=dist)
    tr = BatchTracker(*scene_arrays(sc), W, H, T)
    for kind in ("dark", "noise20"):
        if kind == "dark":
            fr = torch.zeros((C * T, H, W), dtype=torch.uint8, device="cuda")
        else:
            fr = torch.randint(0, 20, (C * T, H, W), dtype=torch.uint8, device="cuda")
        for _ in range(3):
            tr.extract(fr)
        torch.cuda.synchronize()
        tr.ctx.profile(True)
        for _ in range(20):
            tr.extract(fr)
        torch.cuda.synchronize()
        tr.ctx.profile(False)
        p = tr.ctx.profile_read()
        print(name, kind, {k: round(p[k + "_ms"] / max(1, p[k + "_launches"]), 4) for k in ("scan", "patch", "filter", "contour")}, tr.ctx.tile_stats(), flush=True)
    del tr
