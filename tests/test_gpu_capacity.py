"""Nothing is shortened silently (VERDICT r01 "What's weak" 7): the batched path fails a time step
whose point lists do not fit, exactly where the reference -- which has no capacity limits (lib/Helpers.py:191,
203-245) -- would have used every point.  Through the C-ABI on the GPU."""
import numpy as np
import pytest

import oracle
from mocapv2_amd.synth import ZERO_DIST, Scene

pytestmark = pytest.mark.gpu


def disc_frame(H, W, centres, r=15.0):
    img = np.zeros((H, W), np.uint8)
    for u, v in centres:
        y0, y1, x0, x1 = int(v - r - 2), int(v + r + 3), int(u - r - 2), int(u + r + 3)
        yy, xx = np.mgrid[y0:y1, x0:x1]
        d = np.sqrt((xx - u) ** 2 + (yy - v) ** 2)
        np.maximum(img[y0:y1, x0:x1], (np.clip((r + 0.75 - d) / 1.5, 0, 1) * 255).astype(np.uint8), out=img[y0:y1, x0:x1])
    return img


def grid_centres(nx, ny, W, H):
    return [(W * (i + 0.5) / nx, H * (j + 0.5) / ny) for j in range(ny) for i in range(nx)]


def trackers(scene, T, max_points):
    from mocapv2_amd.pipeline import BatchTracker, scene_arrays
    arrays = scene_arrays(scene)
    return BatchTracker(*arrays, scene.width, scene.height, T, max_points=max_points), arrays


def test_more_blobs_than_max_points_fails_the_time_step():
    """40 blobs in camera 1 of time step 1 at max_points = 32: the blob record carries the true count (40), the
    correspondence kernel reports MOCAP_CORR_E_TRUNCATED for that time step only, BatchTracker.finish and
    ReplayTracker raise; with max_points = 64 the same frames go through and match the oracle."""
    import torch
    from mocapv2_amd.pipeline import CapacityError
    from mocapv2_amd.replay import ReplayTracker
    C, T, H, W = 2, 3, 540, 960
    sc = Scene(C, W, H, dist=ZERO_DIST)
    frames = sc.render_batch(seed=11, n_steps=T, n_markers=3, radius_range=(15, 18), noise_max=20)
    frames[1, 1] = disc_frame(H, W, grid_centres(8, 5, W, H))
    dev = torch.from_numpy(frames.reshape(T * C, H, W)).cuda()
    tr, arrays = trackers(sc, T, 32)
    out = tr.step(dev)
    torch.cuda.synchronize()
    cnt = tr.records[:, 0].cpu().numpy().reshape(T, C)
    assert cnt[1, 1] == 40 and cnt[0].max() <= 32
    n = out["n"].cpu().numpy()
    assert n[1] == -3 and n[0] >= 0 and n[2] >= 0
    with pytest.raises(CapacityError) as e:
        tr.finish(out)
    assert e.value.step == 1 and e.value.code == -3 and "max_points" in str(e.value)
    rp = ReplayTracker(*arrays, W, H, batch=2, max_points=32)
    with pytest.raises(CapacityError) as e:
        list(rp.run(frames))
    assert e.value.step == 1
    # enough room: same frames, every time step equals the oracle's
    tr64, _ = trackers(sc, T, 64)
    out = tr64.step(dev)
    n = tr64.finish(out)
    K, dist, R, t, F = arrays
    for s in range(T):
        lists = [oracle.find_dot(frames[s, c], K[c], dist[c]) for c in range(C)]
        P = max(1, max(len(l) for l in lists))
        pts = np.zeros((C, P, 2))
        counts = np.array([len(l) for l in lists], np.int32)
        for c, l in enumerate(lists):
            if l:
                pts[c, :len(l)] = l
        ref = oracle.correspond(pts, counts, K, dist, R, t, F)
        assert n[s] == len(ref["root"])
        assert np.array_equal(out["grp"][s, :n[s]].cpu().numpy(), ref["groups"])


def test_blob_stage_capacity_error_reaches_the_caller():
    """A frame full of small squares (3600 borders > the contour kernel's 384) makes the blob stage report a negative
    count; the correspondence kernel turns that into MOCAP_CORR_E_BLOB for the time step, the trackers raise, and the
    drop-in _find_dot raises as before."""
    import torch
    from mocapv2_amd.pipeline import CapacityError
    C, T, H, W = 2, 2, 540, 960
    sc = Scene(C, W, H, dist=ZERO_DIST)
    frames = sc.render_batch(seed=5, n_steps=T, n_markers=3, radius_range=(15, 18), noise_max=20)
    busy = np.zeros((H, W), np.uint8)
    for y in range(6, H - 12, 24):
        for x in range(6, W - 12, 24):
            busy[y:y + 12, x:x + 12] = 255
    frames[0, 0] = busy
    tr, arrays = trackers(sc, T, 32)
    out = tr.step(torch.from_numpy(frames.reshape(T * C, H, W)).cuda())
    torch.cuda.synchronize()
    cnt = tr.records[:, 0].cpu().numpy().reshape(T, C)
    assert cnt[0, 0] < 0, cnt
    n = out["n"].cpu().numpy()
    assert n[0] == -4 and n[1] >= 0
    with pytest.raises(CapacityError) as e:
        tr.finish(out)
    assert e.value.step == 0 and e.value.code == -4
    # camera 1 failing (not the root camera) is caught just the same
    frames[0, 0], frames[0, 1] = frames[1, 0].copy(), busy
    out = tr.step(torch.from_numpy(frames.reshape(T * C, H, W)).cuda())
    with pytest.raises(CapacityError):
        tr.finish(out)


def test_dense_correspond_rejects_counts_beyond_capacity():
    """mocap_correspond on a dense [T][C][P][2] array: a count above P or below 0 fails that time step with its own code."""
    import torch
    from mocapv2_amd.engine import MocapContext
    C, P, T = 3, 8, 4
    sc = Scene(C, dist=ZERO_DIST)
    rng = np.random.default_rng(2)
    ctx = MocapContext(1, 1)
    ctx.set_cameras(np.stack([sc.K] * C), np.stack([sc.dist] * C), np.stack([p["R"] for p in sc.poses]),
                    np.stack([p["t"] for p in sc.poses]))
    ctx.set_fundamentals(np.stack(sc.Fs))
    pts = np.zeros((T, C, P, 2), np.int32)
    cnt = np.zeros((T, C), np.int32)
    for s in range(T):
        cents = sc.centroids(sc.markers(rng, 5))
        for c in range(C):
            pts[s, c, :5] = cents[c]
            cnt[s, c] = 5
    cnt[1, 2] = P + 1
    cnt[2, 0] = -3
    cnt[3, 1] = P  # exactly full is fine (slots 5.. hold zeros: they are points like any other)
    out = ctx.correspond(torch.from_numpy(pts).cuda(), torch.from_numpy(cnt).cuda())
    n = out["n"].cpu().numpy()
    assert n[0] == 5 and n[1] == -3 and n[2] == -4 and n[3] >= 0


def test_many_roots_of_moderate_group_counts_and_the_step_budget():
    """A time step whose roots each stay far below max_groups but whose groups add up beyond the default per-step share of the
    error scratch (max(2 * max_groups, 8192)): reported as MOCAP_CORR_E_GROUPS with the default, answered -- equal to the
    oracle, which has no such limit (like the reference, lib/Helpers.py:239-245) -- once corr_step_groups is raised."""
    import torch
    import oracle
    from mocapv2_amd.engine import MocapContext
    C, P = 4, 24
    sc = Scene(C, dist=ZERO_DIST)
    K, dist = np.stack([sc.K] * C), np.stack([sc.dist] * C)
    R, t = np.stack([p["R"] for p in sc.poses]), np.stack([p["t"] for p in sc.poses])
    F = np.stack(sc.Fs)
    # three groups of 8 camera-0 roots a few pixels apart; in every other camera 8 points along each group's epipolar line:
    # every root sees 8 candidates per camera -> 8^3 = 512 groups (far below max_groups), 24 roots -> 12 288 in the time step
    pts = np.zeros((1, C, P, 2), np.int32)
    cnt = np.full((1, C), P, np.int32)
    for g in range(3):
        for k in range(8):
            pts[0, 0, 8 * g + k] = (500 + 400 * g + k // 2, 300 + 200 * g + k % 2)
    for i in range(1, C):
        for g in range(3):
            a, b, c = oracle.epiline(F[i - 1], *pts[0, 0, 8 * g + 3]).astype(float)
            d = a * 960 + b * 540 + c
            for m in range(8):
                pts[0, i, 8 * g + m] = np.round([960 - a * d - b * 40 * (m - 3.5), 540 - b * d + a * 40 * (m - 3.5)])
    ref = oracle.correspond(pts[0].astype(float), cnt[0], K, dist, R, t, F)
    ctx = MocapContext(1, 1)
    ctx.set_cameras(K, dist, R, t)
    ctx.set_fundamentals(F)
    d_pts, d_cnt = torch.from_numpy(pts).cuda(), torch.from_numpy(cnt).cuda()
    n = int(ctx.correspond(d_pts, d_cnt, max_groups=2048)["n"].cpu()[0])
    assert n == -2  # every root below max_groups, the step's total above max(2 * 2048, 8192)
    ctx.set_tuning("corr_step_groups", 1 << 18)
    out = {k: v.cpu().numpy() for k, v in ctx.correspond(d_pts, d_cnt, max_groups=2048).items()}
    k = int(out["n"][0])
    assert k == len(ref["root"]) and k > 0
    assert np.array_equal(out["grp"][0, :k], ref["groups"]) and np.abs(out["xyz"][0, :k] - ref["xyz"]).max() < 1e-7
    assert np.array_equal(out["order"][0, :k], ref["order"])


@pytest.mark.parametrize("C,P,M", [(16, 128, 40), (6, 255, 60), (32, 255, 12), (16, 96, 30)], ids=["c16_p128", "c6_p255", "c32_p255", "c16_p96"])
def test_large_point_capacities_fit_the_lds_plan(C, P, M):
    """the correspondence kernel's LDS plan -- camera block, candidate lists, staged points, per-group errors --
    must adapt to P points x C cameras instead of rejecting sizes the first plan accepted: P = 128 x C = 16 is what
    bench.py --cameras 16 --markers 64 (BASELINE configs[4]) asks for, P = 255 is the documented maximum.  The staged points
    and the in-LDS errors give way first (both have fallbacks); the results equal the oracle's."""
    import torch
    from mocapv2_amd.engine import MocapContext
    sc = Scene(C, 3840, 2160, dist=ZERO_DIST, radius=4.0)
    K, dist = np.stack([sc.K] * C), np.stack([sc.dist] * C)
    R, t, F = np.stack([p["R"] for p in sc.poses]), np.stack([p["t"] for p in sc.poses]), np.stack(sc.Fs)
    rng = np.random.default_rng(100 * C + P)
    T = 2
    pts = np.zeros((T, C, P, 2), np.int32)
    cnt = np.zeros((T, C), np.int32)
    for s in range(T):
        cents = sc.centroids(sc.markers(rng, M, extent=1.2), rng, jitter=0.3)
        for c in range(C):
            k = M if (s + c) % 3 else M - 2  # ragged counts
            pts[s, c, :k] = cents[c][rng.permutation(M)[:k]]
            cnt[s, c] = k
    ctx = MocapContext(1, 1)
    ctx.set_cameras(K, dist, R, t)
    ctx.set_fundamentals(F)
    out = {k: v.cpu().numpy() for k, v in ctx.correspond(torch.from_numpy(pts).cuda(), torch.from_numpy(cnt).cuda(), max_groups=1 << 16).items()}
    total = 0
    for s in range(T):
        n = int(out["n"][s])
        try:
            ref = oracle.correspond(pts[s].astype(float), cnt[s], K, dist, R, t, F, max_groups=1 << 16)
        except RuntimeError:  # a root with more than max_groups candidate groups: given up by both
            assert n == -2
            continue
        assert n == len(ref["root"])
        if n:
            assert np.array_equal(out["grp"][s, :n], ref["groups"])
            assert np.abs(out["xyz"][s, :n] - ref["xyz"]).max() < 1e-7
            assert np.array_equal(out["order"][s, :n], ref["order"])
        total += n
    assert total >= 1  # (a root needs a candidate in every other camera: few survive 31 of them)
