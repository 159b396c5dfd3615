"""A/B of north_star's "frames shard one-camera-per-stream on one GPU" against the product's one time-major launch over all
cameras (reference RealtimeTracking_FLIR.py:304-312: one thread per camera).  Same frames, same box:
  (a) BatchTracker, depth 1 and depth 3: every kernel of stage A is ONE launch over all C x T images;
  (b) one MocapContext + one HIP stream per camera: stage A as C concurrent launch chains over T images each (strided views of
      the same time-major frames), joined by events, then one correspondence launch over camera-major records.
Results of (b) are checked against (a).   python scratch/ab_camera_streams.py [--markers 8] [--steps 20]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Workload  # noqa: E402
from mocapv2_amd.engine import MocapContext  # noqa: E402
from mocapv2_amd.pipeline import BatchTracker, scene_arrays  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--markers", type=int, default=8)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--time-steps", type=int, default=512)
args = ap.parse_args()
C, W, H, T = 6, 1920, 1080, args.time_steps
wl = Workload(C, W, H, args.markers, "mild")
scene = wl.scene()
K, dist, R, t, F = scene_arrays(scene)
host = wl.render(scene, [(c, s) for s in range(32) for c in range(C)])  # 32 rendered time steps, repeated
reps = T // 32
frames = torch.from_numpy(host).cuda().reshape(32, C, H, W).repeat(reps, 1, 1, 1).reshape(T * C, H, W).contiguous()
torch.cuda.synchronize()


def timed(fn, sync, steps):
    for _ in range(3):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    return (time.perf_counter() - t0) / steps * 1e3


res = {}
for depth in (1, 3):
    trk = BatchTracker(K, dist, R, t, F, W, H, T, max_points=wl.max_points, max_groups=wl.max_groups, depth=depth)
    res[f"time_major_depth{depth}_ms"] = timed(lambda: trk.step(frames), trk.synchronize, args.steps)
    if depth == 1:
        ref_out = {k: v.clone() for k, v in trk.step(frames).items()}
        trk.synchronize()
        ref_rec = trk.records.clone()
    del trk
torch.cuda.empty_cache()

# (b) one context + one stream per camera
ctxs, streams, recs = [], [], torch.zeros((C, T, 2 + 2 * wl.max_points), dtype=torch.int32, device="cuda")
for c in range(C):
    ctx = MocapContext(W, H, 1)
    ctx.set_undistort(0, K[c], dist[c])
    ctxs.append(ctx)
    streams.append(torch.cuda.Stream())
main = MocapContext(W, H, 1)
main.set_cameras(K, dist, R, t)
main.set_fundamentals(F)
views = [frames.reshape(T, C, H, W)[:, c] for c in range(C)]
out = None


def per_camera():
    global out
    cur = torch.cuda.current_stream()
    for c in range(C):
        streams[c].wait_stream(cur)
        with torch.cuda.stream(streams[c]):
            ctxs[c].blob_centroids(views[c], cam_mod=1, max_blobs=wl.max_points, records=recs[c])
    for c in range(C):
        cur.wait_stream(streams[c])
    out = main.correspond_records(recs, T, C, stride_t=1, stride_c=T, P=wl.max_points, max_groups=wl.max_groups, out=out)


res["camera_streams_ms"] = timed(per_camera, torch.cuda.synchronize, args.steps)
per_camera()
torch.cuda.synchronize()
same = bool(torch.equal(out["n"], ref_out["n"]))
n = ref_out["n"].clamp(min=0)
live = torch.arange(out["xyz"].shape[1], device="cuda")[None, :] < n[:, None]
same = same and bool(torch.equal(out["xyz"][live], ref_out["xyz"][live])) and bool(torch.equal(out["grp"][live], ref_out["grp"][live]))
cnt_a = ref_rec[:, 0].reshape(T, C)
same = same and bool(torch.equal(recs[:, :, 0].t().contiguous(), cnt_a))
res["camera_streams_same_results"] = same
res["frames_per_s"] = {k[:-3]: round(T / v * 1e3, 1) for k, v in res.items() if k.endswith("_ms")}
res["config"] = f"{C} x {W}x{H}, {args.markers} markers, {T} time steps per batch"
print(res)
