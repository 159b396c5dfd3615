"""Parity and size-independent properties at BASELINE.json's full sizes (1080p / 4K, 6-16 cameras, 8-64 markers)."""
import numpy as np
import pytest

import oracle
from mocapv2_amd.synth import MILD_DIST, ZERO_DIST, Scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dist", [ZERO_DIST, MILD_DIST], ids=["nodist", "mild"])
def test_1080p_six_camera_frames_match_oracle(dist):
    """BASELINE configs[1]: 6 x 1920x1080, 8 markers -- every image's centroid list equals the oracle's."""
    import torch
    from mocapv2_amd.engine import MocapContext
    sc = Scene(6, 1920, 1080, dist=dist)
    frames = sc.render_batch(seed=2000, n_steps=2, n_markers=8, radius_range=(16, 22), salt=0.001)  # [2,6,H,W]
    ctx = MocapContext(1920, 1080, n_slots=6)
    for s in range(6):
        ctx.set_undistort(s, sc.K, sc.dist)
    xy, cnt = ctx.record_views(ctx.blob_centroids(torch.from_numpy(frames).cuda(), cam_mod=6))
    xy, cnt = xy.cpu().numpy(), cnt.cpu().numpy()
    flat = frames.reshape(-1, 1080, 1920)
    n_pts = 0
    for i in range(len(flat)):
        exp = oracle.find_dot(flat[i], sc.K, sc.dist)
        assert cnt[i] == len(exp) and xy[i, :cnt[i]].tolist() == exp, i
        n_pts += len(exp)
    assert n_pts >= 60


def test_4k_frame_64_markers_matches_oracle():
    """BASELINE configs[4] geometry: 3840x2160, 64 markers."""
    import torch
    from mocapv2_amd.engine import MocapContext
    sc = Scene(16, 3840, 2160, dist=MILD_DIST)
    rng = np.random.default_rng(77)
    mk = sc.markers(rng, 64, extent=1.0)
    img = sc.render(rng, mk, 3, radius_range=(18, 26), salt=0.0005)
    ctx = MocapContext(3840, 2160, 1)
    ctx.set_undistort(0, sc.K, sc.dist)
    xy, cnt = ctx.record_views(ctx.blob_centroids(torch.from_numpy(img).cuda()))
    exp = oracle.find_dot(img, sc.K, sc.dist)
    n = int(cnt.cpu()[0])
    assert n == len(exp) and xy[0, :n].cpu().numpy().tolist() == exp and n >= 40


def test_batch_invariance_and_remap_variants_agree_at_1080p(monkeypatch):
    """Size-independent properties: an image's result does not depend on the batch it travels in, nor on which of the
    remap variants (general dense kernel, box kernel staged / unstaged) produced it."""
    import torch
    from mocapv2_amd.engine import MocapContext
    sc = Scene(6, 1920, 1080, dist=MILD_DIST)
    frames = sc.render_batch(seed=3000, n_steps=4, n_markers=8, radius_range=(16, 22), salt=0.001)
    dev = torch.from_numpy(frames).cuda()
    results = []
    for mode in ["general", "3", "box_unstaged"]:  # "3" = the product path (box kernel, staged source)
        monkeypatch.delenv("MOCAP_GENERAL_FILTER", raising=False)
        monkeypatch.delenv("MOCAP_BOX_STAGE_BYTES", raising=False)
        if mode == "general":
            monkeypatch.setenv("MOCAP_GENERAL_FILTER", "1")
        if mode == "box_unstaged":
            monkeypatch.setenv("MOCAP_BOX_STAGE_BYTES", "0")
        ctx = MocapContext(1920, 1080, n_slots=6)
        for s in range(6):
            ctx.set_undistort(s, sc.K, sc.dist)
        mask = ctx.filter_mask(dev, cam_mod=6).cpu().numpy()
        rec = ctx.blob_centroids(dev, cam_mod=6).cpu().numpy()
        results.append((mask, rec))
        if mode == "3":  # one image at a time == inside the batch
            for t in range(4):
                for c in range(6):
                    one = ctx.blob_centroids(dev[t, c], cam_mod=1, slot_base=c).cpu().numpy()[0]
                    n = one[0]
                    assert n == rec[t * 6 + c][0] and np.array_equal(one[2:2 + 2 * n], rec[t * 6 + c][2:2 + 2 * n])
    for mask, rec in results[1:]:
        assert np.array_equal(mask, results[0][0])
        cnt = rec[:, 0]
        assert np.array_equal(cnt, results[0][1][:, 0])
        for i, n in enumerate(cnt):
            assert np.array_equal(rec[i, 2:2 + 2 * n], results[0][1][i, 2:2 + 2 * n])


def test_config5_sixteen_cameras_64_markers_and_ba_residuals():
    """BASELINE configs[4]: 16 cameras, 64 markers (spread so that the reference's cartesian expansion stays finite:
    a denser cloud makes single roots exceed 4M candidate groups, which both the oracle and the kernel report as
    MOCAP_CORR_E_GROUPS): correspondence + DLT against the oracle, and the bundle-adjustment
    residual vector (triangulate + reproject per point, float32) for a 16-camera parameter vector."""
    import torch
    from mocapv2_amd.engine import MocapContext
    C, M = 16, 64
    sc = Scene(C, 3840, 2160, dist=MILD_DIST, radius=4.0)
    K, dist = np.stack([sc.K] * C), np.stack([sc.dist] * C)
    R, t, F = np.stack([p["R"] for p in sc.poses]), np.stack([p["t"] for p in sc.poses]), np.stack(sc.Fs)
    rng = np.random.default_rng(5)
    cents = sc.centroids(sc.markers(rng, M, extent=1.2), rng, jitter=0.3)
    pts = np.zeros((1, C, M, 2), np.int32)
    cnt = np.full((1, C), M, np.int32)
    for c in range(C):
        pts[0, c] = cents[c][rng.permutation(M)]
    ctx = MocapContext(1, 1)
    ctx.set_cameras(K, dist, R, t)
    ctx.set_fundamentals(F)
    out = ctx.correspond(torch.from_numpy(pts).cuda(), torch.from_numpy(cnt).cuda(), max_groups=1 << 22)
    ref = oracle.correspond(pts[0].astype(float), cnt[0], K, dist, R, t, F)
    n = int(out["n"].cpu()[0])
    assert n == len(ref["root"]) and n >= 32
    assert np.array_equal(out["grp"][0, :n].cpu().numpy(), ref["groups"])
    assert np.abs(out["xyz"][0, :n].cpu().numpy() - ref["xyz"]).max() < 1e-7
    assert np.array_equal(out["order"][0, :n].cpu().numpy(), ref["order"])

    # BA residual vector: camera 0 at the origin, cameras 1.. as (rotvec, t) relative poses
    from scipy.spatial.transform import Rotation
    R0, t0 = R[0], t[0]
    params = []
    for c in range(1, C):
        Rrel = R[c] @ R0.T
        trel = t[c] - Rrel @ t0
        params += list(Rotation.from_matrix(Rrel).as_rotvec()) + list(trel)
    params = np.array(params) + rng.normal(0, 1e-3, 6 * (C - 1))
    groups = ref["groups"]                                 # [n, C, 2] matched image points
    valid = np.ones(groups.shape[:2], np.uint8)
    exp = oracle.ba_residuals(params, C, groups, valid, K, dist)
    import mocapv2_amd.lib.Helpers as H
    H.camera_params = np.array([{"intrinsic_matrix": K[i].tolist(), "distortion_coef": dist[i].tolist()} for i in range(C)])
    poses = H.params_to_camera_poses(params, C)
    obj = H.triangulate_points(groups.tolist(), poses)
    got = H.calculate_reprojection_errors(groups.tolist(), obj, poses).astype(np.float32)
    assert got.shape == exp.shape and np.allclose(got, exp, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("markers", [8, 32], ids=["configs1", "configs2"])
def test_full_batch_512_time_steps_properties_and_oracle_sample(markers):
    """BASELINE.json configs[1] / configs[2] (8 / 32 markers) at full size -- ONE batch of 512 time steps x 6 cameras x 1080p (3072 images, 6.4 GB) through
    BatchTracker -- checked by what does not depend on the size: (a) the batch holds 16 copies of 32 rendered time steps: every copy of
    a time step gives the same centroid record and the same 3-D points; (b) permuting the time steps of the batch permutes the results
    and changes nothing else (a result does not depend on its neighbours in the batch, on the wave / workgroup / list position its blobs
    land in, nor on the lane of the pipeline); (c) eight of the 32 time steps against the CPU oracle, frame -> centroid -> 3-D."""
    import torch
    from mocapv2_amd.pipeline import BatchTracker, scene_arrays
    from test_gpu_oracle_e2e import assert_records_equal, assert_step_equal, oracle_step
    C, T, W, H, base = 6, 512, 1920, 1080, 32
    sc = Scene(C, W, H, dist=MILD_DIST)
    K, dist, R, t, F = scene_arrays(sc)
    frames = sc.render_batch(seed=5100, n_steps=base, n_markers=markers, radius_range=(16, 22), salt=0.001)  # [32, 6, H, W]
    max_groups = 4096  # a time step whose cartesian expansion exceeds it is given up by kernel and oracle alike (n = -2)
    dev32 = torch.from_numpy(frames).cuda()
    g = torch.Generator().manual_seed(5)
    order = torch.cat([torch.randperm(base, generator=g) for _ in range(T // base)])  # time step s of the batch = rendered step order[s]
    batch = dev32[order.cuda()].reshape(T * C, H, W).contiguous()
    perm = torch.randperm(T, generator=g)
    batch_p = batch.reshape(T, C, H, W)[perm.cuda()].reshape(T * C, H, W).contiguous()
    trk = BatchTracker(K, dist, R, t, F, W, H, T, depth=3, max_points=2 * markers if markers > 16 else 32, max_groups=max_groups)

    def run(x):
        out = trk.step(x)
        trk.synchronize()
        rec = trk.records.cpu().numpy().copy()
        return rec, {k: v.cpu().numpy().copy() for k, v in out.items()}

    rec, out = run(batch)           # lane 0
    rec_p, out_p = run(batch_p)     # lane 1
    rec2, out2 = run(batch)         # lane 2
    def same_record(a, b):
        return a[0] == b[0] and np.array_equal(a[2:2 + 2 * max(0, a[0])], b[2:2 + 2 * max(0, a[0])])

    assert all(same_record(rec[i], rec2[i]) for i in range(T * C)) and np.array_equal(out["n"], out2["n"])  # another lane, same results
    order_np, perm_np = order.numpy(), perm.numpy()
    first = {}
    points = 0
    for s in range(T):
        k = int(out["n"][s])
        assert k >= 0 or k == -2, (s, k)
        n_s = k
        k = max(k, 0)
        live = [(rec[s * C + c, 0], rec[s * C + c, 2:2 + 2 * max(0, rec[s * C + c, 0])].tolist()) for c in range(C)]
        res = (n_s, out["root"][s, :k].tolist(), out["grp"][s, :k].tolist(), out["xyz"][s, :k].tolist(), out["order"][s, :k].tolist())
        r = int(order_np[s])
        if r in first:
            assert (live, res) == first[r], ("copies of one time step differ", s, r)                       # (a)
        else:
            first[r] = (live, res)
        points += k
    for j in range(T):  # (b): position j of the permuted batch holds time step perm[j]
        s = int(perm_np[j])
        for c in range(C):
            assert same_record(rec_p[j * C + c], rec[s * C + c]), (j, s, c)
        k = max(int(out_p["n"][j]), 0)
        assert int(out_p["n"][j]) == int(out["n"][s]) and np.array_equal(out_p["xyz"][j, :k], out["xyz"][s, :k]) and np.array_equal(out_p["grp"][j, :k], out["grp"][s, :k]), (j, s)
    assert len(first) == base and points >= 3 * T  # (32 markers: 8-12 triangulated roots per time step that is not given up)
    for r in range(0, base, 4):  # (c)
        s = int(np.nonzero(order_np == r)[0][0])
        lists, ref = oracle_step(frames[r], K, dist, R, t, F, max_groups=max_groups)
        for c in range(C):
            assert_records_equal(rec, s * C + c, lists[c], (s, c))
        if ref is None:
            assert out["n"][s] == -2, s
        else:
            assert_step_equal(out, s, ref, s)
