import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mocapv2_amd.engine import MocapContext, REC_INTS  # noqa
from mocapv2_amd.pipeline import scene_arrays
from mocapv2_amd.synth import MILD_DIST, Scene
W, H, C, T = 1920, 1080, 6, 64
sc = Scene(C, W, H, dist=MILD_DIST)
K, D, R, t, F = scene_arrays(sc)
frames = np.empty((T, C, H, W), np.uint8)
for s in range(T):
    rng = np.random.default_rng(100 + s)
    mk = sc.markers(rng, 8)
    for c in range(C):
        frames[s, c] = sc.render(np.random.default_rng(7 * s + c), mk, c)
fr = torch.from_numpy(frames).cuda()
cx = MocapContext(W, H, C)
for c in range(C):
    cx.set_undistort(c, K[c], D[c])
for _ in range(4):
    cx.blob_centroids(fr, cam_mod=C)
torch.cuda.synchronize()
