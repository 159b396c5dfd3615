"""Soak: the same resident batch through the three-lane pipeline many times; every lane must end with identical records
and 3-D outputs (a race between batches in flight would show as a difference).  python scratch/soak_determinism.py [steps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocapv2_amd.pipeline import BatchTracker, scene_arrays  # noqa: E402
from mocapv2_amd.synth import MILD_DIST, Scene  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
T, C, W, H = 128, 6, 1920, 1080
sc = Scene(C, W, H, dist=MILD_DIST)
frames = torch.from_numpy(sc.render_batch(5000, T, 8, radius_range=(16.0, 22.0), salt=0.001).reshape(T * C, H, W)).cuda()
bt = BatchTracker(*scene_arrays(sc), W, H, T, depth=3)
ref = None
for rounds in range(steps // 3):
    for _ in range(3):
        bt.step(frames)
    if rounds % 10 == 9 or rounds == steps // 3 - 1:
        bt.synchronize()
        for lane in bt.lanes:
            cur = (lane.records.clone(), lane.out["n"].clone(), lane.out["xyz"].clone())
            if ref is None:
                ref = cur
            n = ref[1]
            assert torch.equal(cur[0], ref[0]) and torch.equal(cur[1], n), f"records differ after {3 * (rounds + 1)} steps"
            k = int(n.max())
            assert torch.equal(cur[2][:, :k][n[:, None] > torch.arange(k, device=n.device)[None, :]],
                               ref[2][:, :k][n[:, None] > torch.arange(k, device=n.device)[None, :]])
print(f"{steps} steps, 3 lanes: identical records and 3-D points throughout; points per frame {float(ref[1].float().mean()):.2f}")
