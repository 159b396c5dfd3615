"""Calibration-time batch paths: the arithmetic of the reference's
`CalculateCameraPoses.py` -- relative pose from a fundamental matrix with the four-candidate cheirality vote
(:195-231), bundle adjustment (lib/Helpers.py:158-176), origin / floor alignment (:283-361) and
`poses_to_fundamental_matrix` (:26-78) -- on top of the same HIP triangulation / reprojection kernels the per-frame
path uses.

What is batched that the reference loops over:
  * the four (R, t) candidates are triangulated by ONE launch over 4 N groups (the reference calls
    `triangulate_points` four times, one Python SVD per point);
  * one bundle-adjustment Jacobian (SciPy's 2-point forward differences: 6 perturbed parameter vectors per free
    camera) is ONE triangulation launch + ONE reprojection launch over 7 N groups instead of 6 x 2 x N Python calls;
    steps and rounding are SciPy's, so the optimiser walks the same iterates as the reference's
    `least_squares(..., jac='2-point')`.
Image capture, `cv.findFundamentalMat` (RANSAC) and the plotting / JSON writing of that script are outside the path;
fundamental matrices come in as arguments.
"""
import json
from itertools import combinations

import numpy as np

from .engine import default_context


# ---- small closed forms -----------------------------------------------------------------------------------------
def poses_to_fundamental_matrix(pose1, pose2, K1=None, K2=None):
    """reference CalculateCameraPoses.py:26-78: F with x2^T F x1 = 0 from two world->camera poses; the essential
    matrix when no intrinsics are given."""
    R1, t1 = np.asarray(pose1["R"], float), np.asarray(pose1["t"], float).reshape(3, 1)
    R2, t2 = np.asarray(pose2["R"], float), np.asarray(pose2["t"], float).reshape(3, 1)
    R_rel = R2 @ R1.T
    t_rel = (t2 - R2 @ R1.T @ t1).ravel()
    E = np.array([[0, -t_rel[2], t_rel[1]], [t_rel[2], 0, -t_rel[0]], [-t_rel[1], t_rel[0], 0]]) @ R_rel
    if K1 is not None and K2 is not None:
        return np.linalg.inv(np.asarray(K2, float)).T @ E @ np.linalg.inv(np.asarray(K1, float))
    return E


def decompose_essential(E):
    """cv.decomposeEssentialMat (reference :197) in closed form: with E = U diag(1,1,0) Vt, det U = det Vt = +1,
    R1 = U W Vt, R2 = U W^T Vt, t = U[:, 2] (unit length, sign free).  The SVD's sign freedom only permutes the
    candidate set {R1, R2} x {t, -t} that `select_relative_pose` votes over."""
    U, _, Vt = np.linalg.svd(np.asarray(E, float))
    if np.linalg.det(U) < 0:
        U = -U
    if np.linalg.det(Vt) < 0:
        Vt = -Vt
    W = np.array([[0.0, 1.0, 0.0], [-1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    return U @ W @ Vt, U @ W.T @ Vt, U[:, 2].reshape(3, 1)


def _intrinsics(camera_params, n):
    K = [np.asarray(camera_params[i % len(camera_params)]["intrinsic_matrix"], float) for i in range(n)]
    d = [np.asarray(camera_params[i % len(camera_params)]["distortion_coef"], float).ravel()[:5] for i in range(n)]
    return np.array(K), np.array(d)


# ---- relative pose: four candidates, one launch -----------------------------------------------------------------
def select_relative_pose(points1, points2, base_pose, R1, R2, t, camera_params, ctx=None):
    """reference CalculateCameraPoses.py:199-231.  points1/points2 [N, 2] are the same markers seen by the previous
    and the new camera; candidates (R1, t), (R1, -t), (R2, t), (R2, -t) in that order (:201-202).  Every candidate is
    triangulated against `base_pose` (the reference passes `camera_poses[-1]`) with the intrinsics of cameras 0 and 1
    -- the reference indexes them by position in the two-pose list (lib/Helpers.py:59-61) -- and scored by the number
    of points with z > 0 plus the number with (R_i^T X).z > 0 (:214-219, the reference's expression, kept as is).
    The first candidate with the strictly largest score wins (:221).

    Returns dict: index, R, t (the winning candidate), pose (chained onto base_pose, :226-227), counts [4],
    object_points [4, N, 3]."""
    ctx = ctx or default_context()
    p1 = np.asarray(points1, float).reshape(-1, 2)
    p2 = np.asarray(points2, float).reshape(-1, 2)
    N = len(p1)
    assert len(p2) == N and N > 0
    t = np.asarray(t, float).reshape(3)
    cand_R = [np.asarray(R1, float), np.asarray(R1, float), np.asarray(R2, float), np.asarray(R2, float)]
    cand_t = [t, -t, t, -t]
    Rb, tb = np.asarray(base_pose["R"], float).reshape(3, 3), np.asarray(base_pose["t"], float).reshape(3)
    # cameras: 0 = base pose, 1..4 = candidates; group (i, n) sees cameras 0 and 1 + i, so "by position" the
    # intrinsics are those of cameras 0 and 1 for every candidate
    K, d = _intrinsics(camera_params, 2)
    ctx.set_cameras(np.array([K[0]] + [K[1]] * 4), np.array([d[0]] + [d[1]] * 4), np.array([Rb] + cand_R),
                    np.array([tb] + cand_t))
    pts = np.zeros((4 * N, 5, 2))
    valid = np.zeros((4 * N, 5), np.uint8)
    for i in range(4):
        pts[i * N:(i + 1) * N, 0] = p1
        pts[i * N:(i + 1) * N, 1 + i] = p2
        valid[i * N:(i + 1) * N, 0] = 1
        valid[i * N:(i + 1) * N, 1 + i] = 1
    xyz, _ = ctx.triangulate_batch(pts, valid, compact_k=True)
    xyz = xyz.reshape(4, N, 3)
    counts = []
    for i in range(4):
        in_candidate_frame = xyz[i] @ cand_R[i]  # rows = R_i^T X, reference :214
        counts.append(int(np.sum(xyz[i][:, 2] > 0) + np.sum(in_candidate_frame[:, 2] > 0)))
    best, most = None, 0
    for i in range(4):
        if counts[i] > most:
            best, most = i, counts[i]
    if best is None:
        raise ValueError("no candidate places a point in front of a camera (the reference fails here as well: R is None)")
    pose = {"R": cand_R[best] @ Rb, "t": tb.reshape(3, 1) + Rb @ cand_t[best].reshape(3, 1)}
    return {"index": best, "R": cand_R[best], "t": cand_t[best].reshape(3, 1), "pose": pose, "counts": counts,
            "object_points": xyz}


def extrinsics_from_fundamentals(image_points, Fs, camera_params, ctx=None):
    """The pose chain of reference `calculate_extrinsics` (:166-235) once the fundamental matrices are known.
    image_points [C][N][2] (the layout `get_points` returns, :80-89), Fs[i] maps camera i pixels to camera i+1 lines.
    Camera 0 is the identity; E = K2^T F K1 always uses the intrinsics of cameras 0 and 1 (:192-195)."""
    K, _ = _intrinsics(camera_params, 2)
    poses = [{"R": np.eye(3), "t": np.zeros((3, 1))}]
    for i in range(len(image_points) - 1):
        E = K[1].T @ np.asarray(Fs[i], float) @ K[0]
        R1, R2, t = decompose_essential(E)
        poses.append(select_relative_pose(image_points[i], image_points[i + 1], poses[-1], R1, R2, t, camera_params,
                                          ctx)["pose"])
    return poses


# ---- bundle adjustment: residual and Jacobian in batched launches ---------------------------------------------------
def residuals_batched(image_points, param_sets, camera_params, ctx=None, problem=None):
    """Residual vectors (per-point reprojection MSE, float32 -- reference lib/Helpers.py:161-167) of S parameter
    vectors at once: ONE launch (mocap_ba_residuals: rotvec -> R, triangulation, reprojection and the float32 cast on
    the device, one workgroup per parameter vector) over image points that stay resident on the GPU.  Like the
    reference's residual, two cameras are assumed (identity + params[0:6]).  `problem`: the BAProblem of an earlier call
    with the same image points (bundle_adjustment keeps one for the whole optimisation)."""
    if problem is None:
        problem = ba_problem(image_points, camera_params, ctx)
    sets = np.array([np.asarray(p, float)[:6] for p in param_sets])
    return np.stack(problem.residuals(sets))


def ba_problem(image_points, camera_params, ctx=None):
    """Uploads the image points [N, 2, 2] of a two-camera bundle adjustment once (engine.BAProblem) and sets the two
    cameras' intrinsics in the context."""
    ctx = ctx or default_context()
    ip = np.asarray(image_points, float)
    assert ip.ndim == 3 and ip.shape[1:] == (2, 2), "the reference's residual is hard-wired to two cameras (lib/Helpers.py:162)"
    K, d = _intrinsics(camera_params, 2)
    ctx.set_cameras(K, d, np.stack([np.eye(3)] * 2), np.zeros((2, 3)))
    return ctx.ba_problem(ip)


def forward_difference_steps(x0, f_dtype=np.float32):
    """SciPy's default 2-point step (scipy.optimize._numdiff): sqrt(eps) * sign(x) * max(1, |x|) with sign(0) = +1 and
    eps that of the narrower of the parameter and residual dtypes -- float32 for the reference's residuals
    (lib/Helpers.py:165), i.e. a relative step of 3.45e-4."""
    x0 = np.asarray(x0, float)
    eps = np.finfo(np.float64).eps
    if np.issubdtype(f_dtype, np.inexact) and np.dtype(f_dtype).itemsize < 8:
        eps = np.finfo(f_dtype).eps
    sign = (x0 >= 0).astype(float) * 2 - 1
    return eps ** 0.5 * sign * np.maximum(1.0, np.abs(x0))


def residual_and_jacobian(image_points, x0, camera_params, ctx=None, problem=None):
    """(f0, J) with J exactly what `least_squares(jac='2-point')` derives from the float32 residuals: column i =
    (f(x0 + h_i e_i) - f0) / ((x0 + h_i e_i)_i - x0_i), the subtraction done in float32 as NumPy does for the
    reference's float32 residual vectors."""
    x0 = np.asarray(x0, float)
    h = forward_difference_steps(x0)
    sets = [x0]
    for i in range(len(x0)):
        x1 = x0.copy()
        x1[i] = x0[i] + h[i]
        sets.append(x1)
    f = residuals_batched(image_points, sets, camera_params, ctx, problem)
    f0 = f[0]
    J = np.empty((len(f0), len(x0)))
    for i in range(len(x0)):
        dx = sets[i + 1][i] - x0[i]
        J[:, i] = (f[i + 1] - f0) / dx
    return f0, J


def bundle_adjustment(image_points, camera_poses, camera_params, batched_jacobian=True, verbose=0, ctx=None):
    """reference lib/Helpers.py:158-176: trust-region least squares (SciPy 'trf', linear loss, ftol 1e-5, xtol 1e-15)
    over rotvec + translation of camera 1, residual = float32 per-point reprojection MSE of the re-triangulated
    points.  With `batched_jacobian` each Jacobian is two launches (see module docstring); without it SciPy
    differences the batched residual itself -- both walk the same iterates.  Returns the two poses, like
    `params_to_camera_poses(result.x)`."""
    from scipy import optimize
    from scipy.spatial.transform import Rotation

    from .lib.Helpers import params_to_camera_poses

    init = np.array([])
    for pose in camera_poses[1:]:
        init = np.concatenate([init, Rotation.from_matrix(np.asarray(pose["R"], float)).as_rotvec(),
                               np.asarray(pose["t"], float).flatten()])

    problem = ba_problem(image_points, camera_params, ctx)  # image points resident for the whole optimisation

    def fun(x):
        return residuals_batched(image_points, [x], camera_params, ctx, problem)[0]

    def jac(x):
        return residual_and_jacobian(image_points, x, camera_params, ctx, problem)[1]

    result = optimize.least_squares(fun, init, jac=jac if batched_jacobian else "2-point", verbose=verbose, loss="linear",
                                    method="trf", ftol=1e-5, xtol=1e-15)
    return params_to_camera_poses(result.x), result


# ---- origin and floor ---------------------------------------------------------------------------------------------------
def set_origin(camera_poses, points_3d):
    """reference :283-294: subtract the mean of the floor points from every camera's t (in place); returns the
    origin, or None when fewer than three points were given."""
    points_3d = np.asarray(points_3d, float)
    if len(points_3d) > 2:
        origin = np.mean(points_3d, axis=0)
        for pose in camera_poses:
            pose["t"] = pose["t"] - origin
        return origin
    return None


def calculate_normal(points_3d):
    """reference :326-333: unit normal of three points (0 when they are collinear)."""
    if len(points_3d) == 3:
        p = [np.asarray(q, float) for q in points_3d]
        normal = np.cross(p[1] - p[0], p[2] - p[0])
        return 0 if np.linalg.norm(normal) == 0 else normal / np.linalg.norm(normal)
    return np.array([0, 0, 1])


def rotation_matrix_from_vectors(vec_orig, vec_rot):
    """reference :335-363: Rodrigues rotation taking vec_orig onto vec_rot."""
    a = np.asarray(vec_orig, float) / np.linalg.norm(vec_orig)
    b = np.asarray(vec_rot, float) / np.linalg.norm(vec_rot)
    cross = np.cross(a, b)
    cross_norm = np.linalg.norm(cross)
    dot = np.dot(a, b)
    if cross_norm == 0:
        if dot > 0:
            return np.eye(3)
        axis = np.array([1.0, 0.0, 0.0]) if abs(a[0]) < 0.99 else np.array([0.0, 1.0, 0.0])
        cross = np.cross(a, axis)
        cross /= np.linalg.norm(cross)
        cross_norm = 1
    Kx = np.array([[0, -cross[2], cross[1]], [cross[2], 0, -cross[0]], [-cross[1], cross[0], 0]])
    return np.eye(3) + Kx + Kx @ Kx * ((1 - dot) / (cross_norm ** 2))


def set_floor(camera_poses, points_3d):
    """reference :296-324: mean normal over all point triples, rotation taking (0, 0, -1) onto it, every pose
    [R|t] <- Rf^T [R|t] (in place; t becomes a (3, 1) column).  Returns the 3x3 rotation, or None when fewer than
    three points were given."""
    if len(points_3d) <= 2:
        return None
    normals = [calculate_normal([points_3d[a], points_3d[b], points_3d[c]])
               for a, b, c in combinations(range(len(points_3d)), 3)]
    normal = np.mean(normals, axis=0)
    Rf = rotation_matrix_from_vectors(np.array([0, 0, -1]), normal)
    for pose in camera_poses:
        RT = np.eye(4)
        RT[:3, :3] = pose["R"]
        RT[:3, 3] = np.asarray(pose["t"], float).flatten()
        R4 = np.eye(4)
        R4[:3, :3] = Rf
        RT = R4.T @ RT
        pose["R"] = RT[:3, :3]
        pose["t"] = RT[:3, 3].reshape(3, 1)
    return Rf


# ---- stage files: the JSON artefacts the reference's stages hand to each other ----------------------
def get_points(path="./jsons/image_points.json"):
    """reference `get_points` (:80-89): jsons/image_points.json holds [point][camera][2]; returned as
    [camera][point][2]."""
    with open(path) as file:
        image_points = json.load(file)
    return np.transpose(np.array(image_points), (1, 0, 2))


def save_extrinsics(camera_poses, prefix="", directory="./jsons", camera_count=None):
    """reference `save_extrinsics` (:257-273) without its module globals: writes `{directory}/{prefix}extrinsics.json`
    = [{"R": 3x3 list, "t": flat list of 3}, ...], the layout `lib.Helpers.get_extrinsics` reads back (:282-291).
    Returns the file name."""
    n = len(camera_poses) if camera_count is None else camera_count
    extrinsics = []
    for i in range(0, n):
        extrinsics.append({"R": np.asarray(camera_poses[i]["R"]).tolist(),
                           "t": np.asarray(camera_poses[i]["t"]).flatten().tolist()})
    extrinsics_filename = f"{directory}/{prefix}extrinsics.json"
    with open(extrinsics_filename, "w") as outfile:
        json.dump(extrinsics, outfile)
    print("Extrinsics saved to", extrinsics_filename)
    return extrinsics_filename


def save_objects(prefix="", object_points=None, directory="./jsons"):
    """reference `save_objects` (:275-281): `{directory}/{prefix}objects.json` = [[x, y, z], ...]."""
    objects_filename = f"{directory}/{prefix}objects.json"
    with open(objects_filename, "w") as outfile:
        json.dump(np.asarray(object_points).tolist(), outfile)
    print("Object points saved to", objects_filename)
    return objects_filename


def save_fundamentals(pair_Fs, directory="./jsons"):
    """The fundamentals.json dump of reference `calculate_extrinsics` (:189-191,236-240): every camera pair's F is
    appended TWICE (`Fs.append(F.tolist())` twice), so that `lib.Helpers.Fs[i - 1]` of a two-camera rig finds its
    matrix at index 0 and a copy at index 1.  pair_Fs: one 3x3 matrix per consecutive camera pair."""
    Fs = []
    for F in pair_Fs:
        Fs.append(np.asarray(F, float).tolist())
        Fs.append(np.asarray(F, float).tolist())
    filename = f"{directory}/fundamentals.json"
    with open(filename, "w") as outfile:
        json.dump(Fs, outfile)
    return filename


def pair_fundamentals(Fs):
    """Inverse of `save_fundamentals`' doubling: the one-per-pair list `extrinsics_from_fundamentals` expects from
    the list a fundamentals.json written by the reference holds (entries 0, 2, 4, ...)."""
    return [np.asarray(F, float) for F in Fs[::2]]


__all__ = ["poses_to_fundamental_matrix", "decompose_essential", "select_relative_pose", "extrinsics_from_fundamentals",
           "residuals_batched", "forward_difference_steps", "residual_and_jacobian", "bundle_adjustment", "set_origin",
           "calculate_normal", "rotation_matrix_from_vectors", "set_floor", "get_points", "save_extrinsics", "save_objects",
           "save_fundamentals", "pair_fundamentals"]
