"""The scan kernel (+ mark / settle) alone on dark frames and on the bench frames, for the library as built and the environment as set:
python scratch/scan_ab.py [markers ...]   (64 rendered time steps repeated 8 times; not a test, not shipped)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocapv2_amd.pipeline import BatchTracker, scene_arrays  # noqa: E402
from mocapv2_amd.synth import MILD_DIST, Scene  # noqa: E402

W, H, C, T = 1920, 1080, 6, 512
sc = Scene(C, W, H, dist=MILD_DIST)
tr = BatchTracker(*scene_arrays(sc), W, H, T, depth=1, max_points=64)
for kind in ["dark"] + [f"markers{m}" for m in (sys.argv[1:] or ["8"])]:
    if kind == "dark":
        fr = torch.zeros((C * T, H, W), dtype=torch.uint8, device="cuda")
    else:
        part = torch.from_numpy(sc.render_batch(4242, 64, int(kind[7:]), radius_range=(16.0, 22.0), salt=0.001)).cuda()
        fr = part.repeat(T // 64, 1, 1, 1).reshape(T * C, H, W).contiguous()
    for _ in range(3):
        tr.extract(fr)
    torch.cuda.synchronize()
    tr.ctx.profile(True)
    for _ in range(20):
        tr.extract(fr)
    torch.cuda.synchronize()
    tr.ctx.profile(False)
    p = tr.ctx.profile_read()
    print(os.environ.get("TAG", ""), kind, {k[:-3]: round(p[k] / max(1, p[k[:-3] + "_launches"]), 4) for k in p if k.endswith("_ms")}, flush=True)
    del fr
