"""Soak, round 3: three different resident batches (32 markers: refilled walker waves, link walks in the second pass) rotating through the
three-lane pipeline; whatever lane a batch runs on, and whatever ran on that lane before, its records and 3-D outputs must be what the
first pass over it gave.  python scratch/soak_rotating.py [steps] [markers]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocapv2_amd.pipeline import BatchTracker, scene_arrays  # noqa: E402
from mocapv2_amd.synth import MILD_DIST, Scene  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
markers = int(sys.argv[2]) if len(sys.argv) > 2 else 32
T, C, W, H = 96, 6, 1920, 1080
sc = Scene(C, W, H, dist=MILD_DIST)
batches = [torch.from_numpy(sc.render_batch(6000 + 1000 * b, T, markers, radius_range=(16.0, 22.0), salt=0.001).reshape(T * C, H, W)).cuda() for b in range(3)]
bt = BatchTracker(*scene_arrays(sc), W, H, T, depth=3, max_points=2 * markers if markers > 16 else 32)


def valid(rec, out):
    cnt = rec[:, 0]
    live = torch.arange(rec.shape[1] - 2, device=rec.device)[None, :] < 2 * cnt.clamp(min=0)[:, None]
    n = out["n"]
    roots = torch.arange(out["xyz"].shape[1], device=n.device)[None, :] < n.clamp(min=0)[:, None]
    return cnt.clone(), rec[:, 2:][live].clone(), n.clone(), out["xyz"][roots].clone(), out["order"][roots].clone()


ref = {}
checked = 0
for i in range(steps):
    k = (i + i // 3) % 3 if i % 7 else (i // 7) % 3  # a lane sees the batches in changing order
    out = bt.step(batches[k])
    if i < 3 or i % 5 == 0:
        bt.finish(out)
        cur = valid(bt.records, out)
        if k not in ref:
            ref[k] = cur
        for a, b in zip(cur, ref[k]):
            assert a.shape == b.shape and torch.equal(a, b), f"batch {k} differs at step {i}"
        checked += 1
bt.synchronize()
print(f"{steps} steps, 3 lanes, 3 rotating batches of {markers} markers: {checked} checks identical; "
      f"image points per image {float(ref[0][0].float().mean()):.1f}, roots per time step {float(ref[0][2].float().mean()):.1f}")
