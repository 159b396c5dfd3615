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
