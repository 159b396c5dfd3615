import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mocapv2_amd.engine import MocapContext
from mocapv2_amd.pipeline import scene_arrays
from mocapv2_amd.synth import MILD_DIST, ZERO_DIST, Scene
W, H, C = 1920, 1080, 6
sc = Scene(C, W, H, dist=ZERO_DIST)
K, D, R, t, F = scene_arrays(sc)
rng = np.random.default_rng(1000)
mk = sc.markers(rng, 8)
for ncam in (1, 6):
    frames = np.stack([sc.render(np.random.default_rng(1000 * 64 + c), mk, c, radius_range=(16.0, 22.0), salt=0.001) for c in range(ncam)])
    cx = MocapContext(W, H, C)
    for c in range(C):
        cx.set_undistort(c, K[c], D[c])
    fr = torch.from_numpy(frames).cuda()
    for rep in range(3):
        cx.blob_centroids(fr, cam_mod=ncam)
        torch.cuda.synchronize()
        print("ncam", ncam, "rep", rep, cx.tile_stats())
