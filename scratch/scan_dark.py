"""What the scan's marking costs: the scan kernel alone on a full batch (512 time steps x 6 cameras x 1080p) of all-dark frames, of noise
below the excess base, and of the bench frames with 8 / 32 markers (not a test, not shipped).  python scratch/scan_dark.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocapv2_amd.pipeline import BatchTracker, scene_arrays  # noqa: E402
from mocapv2_amd.synth import MILD_DIST, Scene  # noqa: E402

W, H, C, T = 1920, 1080, 6, 512
sc = Scene(C, W, H, dist=MILD_DIST)
tr = BatchTracker(*scene_arrays(sc), W, H, T, depth=1, max_points=64)
for kind in ("dark", "noise40", "markers8", "markers32"):
    if kind == "dark":
        fr = torch.zeros((C * T, H, W), dtype=torch.uint8, device="cuda")
    elif kind == "noise40":
        fr = torch.randint(0, 40, (C * T, H, W), dtype=torch.uint8, device="cuda")
    else:
        m = int(kind[7:])
        fr = torch.from_numpy(sc.render_batch(4242, T, m, radius_range=(16.0, 22.0), salt=0.001).reshape(T * C, H, W)).cuda()
    for _ in range(3):
        tr.extract(fr)
    torch.cuda.synchronize()
    tr.ctx.profile(True)
    for _ in range(10):
        tr.extract(fr)
    torch.cuda.synchronize()
    tr.ctx.profile(False)
    p = tr.ctx.profile_read()
    print(kind, {k[:-3]: round(p[k] / max(1, p[k[:-3] + "_launches"]), 4) for k in p if k.endswith("_ms")}, flush=True)
    del fr
