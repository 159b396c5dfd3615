"""Probe: does running two half-batches on two streams (two contexts) beat one full batch? (not shipped)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mocapv2_amd.engine import MocapContext, REC_INTS
from mocapv2_amd.pipeline import scene_arrays
from mocapv2_amd.synth import MILD_DIST, ZERO_DIST, Scene

W, H, C, T = 1920, 1080, 6, 64
rows = int(os.environ.get("MOCAP_ROWS", "68"))
for name, dist in (("mild", MILD_DIST), ("zero", ZERO_DIST)):
    sc = Scene(C, W, H, dist=dist)
    K, D, R, t, F = scene_arrays(sc)
    frames = np.empty((T, C, H, W), np.uint8)
    for s in range(T):
        rng = np.random.default_rng(100 + s)
        mk = sc.markers(rng, 8)
        for c in range(C):
            frames[s, c] = sc.render(np.random.default_rng(7 * s + c), mk, c)
    fr = torch.from_numpy(frames).cuda()
    for nsplit in (1, 2, 4):
        ctxs = [MocapContext(W, H, C) for _ in range(nsplit)]
        for cx in ctxs:
            for c in range(C):
                cx.set_undistort(c, K[c], D[c])
        streams = [torch.cuda.Stream() for _ in range(nsplit)]
        rec = torch.zeros((T * C, REC_INTS), dtype=torch.int32, device="cuda")
        per = T // nsplit
        def run():
            main = torch.cuda.current_stream()
            for i in range(nsplit):
                streams[i].wait_stream(main)
                with torch.cuda.stream(streams[i]):
                    ctxs[i].blob_centroids(fr[i * per:(i + 1) * per], cam_mod=C, records=rec[i * per * C:(i + 1) * per * C])
            for i in range(nsplit):
                main.wait_stream(streams[i])
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(name, "rows", rows, "split", nsplit, "ms/extract", round(dt * 1e3, 4), "checksum", int(rec[:, 0].sum().item()), flush=True)
        del ctxs
