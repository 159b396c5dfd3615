"""Host-side arithmetic of the calibration-time helpers (mocapv2_amd/calibrate.py) against the
reference-derived fixture tests/golden/calib_cheirality.npz and against properties of the formulas.  No GPU."""
import os

import numpy as np
import pytest

from mocapv2_amd import calibrate as cal
from mocapv2_amd.synth import ZERO_DIST, Scene, fundamental_from_poses, project

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_essential_candidates_contain_the_bundled_pose():
    """jsons/fundamentals.json + camera-params-in.json decompose into candidates one of which is camera 1 of
    jsons/before_ba_extrinsics.json (the reference's own output) to 1e-12."""
    g = np.load(os.path.join(GOLDEN, "calib_cheirality.npz"))
    R1, R2, t = cal.decompose_essential(g["K"][1].T @ g["F"] @ g["K"][0])
    assert t.shape == (3, 1) and abs(np.linalg.norm(t) - 1) < 1e-12
    for R in (R1, R2):
        assert np.abs(R @ R.T - np.eye(3)).max() < 1e-12 and abs(np.linalg.det(R) - 1) < 1e-12
    hits = [(i, s) for i, R in enumerate((R1, R2)) for s in (1, -1)
            if np.abs(R - g["before_ba_R"]).max() < 1e-12 and np.abs(s * t.ravel() - g["before_ba_t"]).max() < 1e-12]
    assert len(hits) == 1
    # the fixture's candidate list is this decomposition up to the SVD's sign freedom
    for R in (R1, R2):
        assert min(np.abs(R - c).max() for c in g["cand_R"]) < 1e-12


def test_fundamental_from_poses_satisfies_the_epipolar_constraint():
    sc = Scene(3, dist=ZERO_DIST)
    rng = np.random.default_rng(3)
    X = sc.markers(rng, 12)
    K = np.asarray(sc.camera_params[0]["intrinsic_matrix"], float)
    for a, b in ((0, 1), (1, 2), (0, 2)):
        F = cal.poses_to_fundamental_matrix(sc.poses[a], sc.poses[b], K, K)
        assert np.allclose(F, fundamental_from_poses(sc.poses[a], sc.poses[b], K, K), rtol=0, atol=1e-18)
        xa = np.c_[project(X, sc.poses[a], K, ZERO_DIST), np.ones(len(X))]
        xb = np.c_[project(X, sc.poses[b], K, ZERO_DIST), np.ones(len(X))]
        assert np.abs(np.einsum("ni,ij,nj->n", xb, F, xa)).max() < 1e-9 * np.abs(F).max() * 1920 * 1920
    # without intrinsics: the essential matrix, whose decomposition holds the relative pose
    E = cal.poses_to_fundamental_matrix(sc.poses[0], sc.poses[1])
    R1, R2, t = cal.decompose_essential(E)
    R_rel = sc.poses[1]["R"] @ sc.poses[0]["R"].T
    t_rel = sc.poses[1]["t"].reshape(3) - R_rel @ sc.poses[0]["t"].reshape(3)
    assert min(np.abs(R - R_rel).max() for R in (R1, R2)) < 1e-12
    assert min(np.abs(s * t.ravel() - t_rel / np.linalg.norm(t_rel)).max() for s in (1, -1)) < 1e-12


@pytest.mark.parametrize("b", [(0.3, -0.2, 0.9), (0, 0, -1), (1, 0, 0), (0.6, 0.0, 0.8)])
def test_rotation_from_vectors(b):
    a = np.array([0.0, 0.0, -1.0])
    R = cal.rotation_matrix_from_vectors(a, np.array(b, float))
    assert np.abs(R @ R.T - np.eye(3)).max() < 1e-12
    assert np.abs(R @ a - np.array(b, float) / np.linalg.norm(b)).max() < 1e-12


def test_rotation_from_opposite_vectors_keeps_the_reference_formula():
    """CalculateCameraPoses.py:347-361: for opposite vectors the reference picks an axis, sets cross_norm = 1 and
    evaluates I + K + 2 K^2 -- the K term belongs to sin(theta) = 0 and makes the result a non-rotation.  Parity
    means the same matrix, not a repaired one."""
    R = cal.rotation_matrix_from_vectors(np.array([0, 0, -1]), np.array([0.0, 0.0, 1.0]))
    assert np.array_equal(R, [[-1, 0, -1], [0, 1, 0], [1, 0, -1]])
    R = cal.rotation_matrix_from_vectors(np.array([1.0, 0, 0]), np.array([-2.0, 0.0, 0.0]))
    assert np.array_equal(R, [[-1, -1, 0], [1, -1, 0], [0, 0, 1]])


def test_set_origin_and_floor():
    rng = np.random.default_rng(5)
    poses = [{"R": np.eye(3), "t": np.zeros(3)}, {"R": Scene(2).poses[1]["R"], "t": np.array([0.5, -0.25, 2.0])}]
    before = [{"R": p["R"].copy(), "t": p["t"].copy()} for p in poses]
    # floor points on a tilted plane, counter-clockwise so every triple has the same normal
    n = np.array([0.1, 0.2, -1.0]) / np.linalg.norm([0.1, 0.2, -1.0])
    u = np.cross(n, [1.0, 0, 0])
    u /= np.linalg.norm(u)
    v = np.cross(n, u)
    ang = np.sort(rng.uniform(0, 2 * np.pi, 5))
    pts = np.array([1.0, 2.0, 3.0]) + np.outer(np.cos(ang), u) + np.outer(np.sin(ang), v)
    assert cal.set_origin(poses, pts[:2]) is None and np.array_equal(poses[1]["t"], before[1]["t"])
    origin = cal.set_origin(poses, pts)
    assert np.allclose(origin, pts.mean(axis=0)) and np.allclose(poses[1]["t"], before[1]["t"] - origin)
    assert cal.set_floor(poses, pts[:2]) is None
    t_mid = [p["t"].copy() for p in poses]
    Rf = cal.set_floor(poses, pts)
    normal = Rf @ np.array([0, 0, -1.0])
    assert abs(abs(normal @ n) - 1) < 1e-12  # the fitted normal is the plane's
    for p, b, tm in zip(poses, before, t_mid):
        assert p["t"].shape == (3, 1)
        assert np.allclose(p["R"], Rf.T @ b["R"]) and np.allclose(p["t"].ravel(), Rf.T @ tm)
    # degenerate triple: the reference's scalar 0 normal
    assert cal.calculate_normal([np.zeros(3), np.ones(3), 2 * np.ones(3)]) == 0
    assert np.array_equal(cal.calculate_normal([np.zeros(3)]), [0, 0, 1])


def test_forward_difference_steps_are_scipys():
    from scipy.optimize import _numdiff
    x = np.array([0.0, -0.0, 3.5, -2.25, 1e-9, -40.0])
    for dt in (np.float32, np.float64):
        h = cal.forward_difference_steps(x, dt)
        eps = np.finfo(dt).eps ** 0.5
        assert np.array_equal(h, eps * np.array([1, 1, 3.5, -2.25, 1, -40.0]))
        if hasattr(_numdiff, "_compute_absolute_step"):
            assert np.array_equal(h, _numdiff._compute_absolute_step(None, x, np.zeros(1, dt), "2-point"))
