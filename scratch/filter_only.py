"""Runs the blob stage a few times on the bench frames (for rocprofv3 passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mocapv2_amd.engine import MocapContext
from mocapv2_amd.pipeline import scene_arrays
from mocapv2_amd.synth import MILD_DIST, ZERO_DIST, Scene
W, H, C, T = 1920, 1080, 6, int(os.environ.get("T", "64"))
sc = Scene(C, W, H, dist=MILD_DIST if os.environ.get("DIST", "mild") == "mild" else ZERO_DIST)
K, D, R, t, F = scene_arrays(sc)
frames = np.empty((T, C, H, W), np.uint8)
for s in range(T):
    mk = sc.markers(np.random.default_rng(1000 + s), 8)
    for c in range(C):
        frames[s, c] = sc.render(np.random.default_rng((1000 + s) * 64 + c), mk, c, radius_range=(16.0, 22.0), salt=0.001)
fr = torch.from_numpy(frames).cuda()
cx = MocapContext(W, H, C)
for c in range(C):
    cx.set_undistort(c, K[c], D[c])
for _ in range(int(os.environ.get("REPS", "4"))):
    cx.blob_centroids(fr, cam_mod=C)
torch.cuda.synchronize()
print("done", cx.tile_stats())
