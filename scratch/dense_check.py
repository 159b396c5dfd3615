"""Early-out off vs on over changing batches on one context: records must be identical (and equal to the oracle's)."""
import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from mocapv2_amd.engine import MocapContext
from mocapv2_amd.synth import MILD_DIST, Scene
W, H, C = 1920, 1080, 2
sc = Scene(C, W, H, dist=MILD_DIST)
batches = [sc.render_batch(seed=100 + 10 * b, n_steps=3, n_markers=8, radius_range=(16, 22), salt=0.001).reshape(3 * C, H, W) for b in range(3)]
ctxs = {}
for name in ("sparse", "dense"):
    ctx = MocapContext(W, H, n_slots=C)
    for s in range(C):
        ctx.set_undistort(s, sc.K, sc.dist)
    if name == "dense":
        ctx.set_tuning("skip_dark", 0)
    ctxs[name] = ctx
for rnd in range(2):
    for b, fr in enumerate(batches):
        dev = torch.from_numpy(fr).cuda()
        recs = {n: c.blob_centroids(dev, cam_mod=C).cpu().numpy() for n, c in ctxs.items()}
        for i in range(len(fr)):
            exp = oracle.find_dot(fr[i], sc.K, sc.dist)
            for n, r in recs.items():
                k = r[i, 0]
                got = r[i, 2:2 + 2 * max(k, 0)].reshape(-1, 2).tolist()
                if k != len(exp) or got != exp:
                    print("MISMATCH", n, "round", rnd, "batch", b, "image", i, "count", k, "expected", len(exp), got[:3], exp[:3])
print("done")
