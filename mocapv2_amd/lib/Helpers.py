"""Mirror of reference lib/Helpers.py: same function names, arguments, return shapes, sentinels and quirks
(listed in SURVEY.md section 8a); triangulation, reprojection and correspondence run in HIP kernels.

Module globals `camera_params`, `camera_params_path`, `Fs` stay assignable, as the reference's callers expect
(reference lib/Helpers.py:9-11).
"""
import json

import numpy as np
import torch

from ..engine import default_context

camera_params = None
camera_params_path = "./jsons/camera-params-in.json"
Fs = []


def get_extrinsics(path="./jsons/after_ba_extrinsics.json"):
    """reference :282-291 (the second definition, which shadows :13-20)."""
    global global_camera_poses
    global camera_count
    with open(path) as file:
        global_camera_poses = json.load(file)
        for i in range(0, len(global_camera_poses)):
            global_camera_poses[i]["R"] = np.array(global_camera_poses[i]["R"])
            global_camera_poses[i]["t"] = np.array(global_camera_poses[i]["t"])
    camera_count = len(global_camera_poses)
    return global_camera_poses, camera_count


def read_fundamental_matrix():
    """reference :22-28"""
    global Fs
    if len(Fs) == 0:
        with open("./jsons/fundamentals.json") as file:
            Fs = json.load(file)
            print("Fundamental matrix loaded")


def read_camera_params():
    """reference :30-40"""
    global camera_params
    if camera_params is None:
        with open(camera_params_path, "r") as file:
            camera_params = np.array(json.load(file))
            print("Camera params loaded")
            return camera_params


# ---- plumbing -----------------------------------------------------------------------------------------------------
def _is_none(p):
    return p[0] is None and p[1] is None


def _sync_cameras(ctx, camera_poses):
    n = len(camera_poses)
    K = np.array([np.asarray(camera_params[i]["intrinsic_matrix"], float) for i in range(n)])
    d = np.array([np.asarray(camera_params[i]["distortion_coef"], float).ravel()[:5] for i in range(n)])
    R = np.array([np.asarray(p["R"], float).reshape(3, 3) for p in camera_poses])
    t = np.array([np.asarray(p["t"], float).reshape(3) for p in camera_poses])
    ctx.set_cameras(K, d, R, t)


def _pack_groups(groups, n_cam):
    """list of groups (each n_cam entries of [x,y] or [None,None]) -> pts [N,C,2] float64, valid [N,C] uint8"""
    N = len(groups)
    pts = np.zeros((N, n_cam, 2))
    valid = np.zeros((N, n_cam), np.uint8)
    for n, g in enumerate(groups):
        for c in range(n_cam):
            p = g[c]
            if not _is_none(p):
                pts[n, c] = (p[0], p[1])
                valid[n, c] = 1
    return pts, valid


def triangulate_point(image_points, camera_poses):
    """reference :43-84.  image_points shape = [camera_count,2]; entries may be [None, None].  Cameras with a None
    entry are dropped and the intrinsics are then taken by position (reference :50-61)."""
    read_camera_params()
    pts, valid = _pack_groups([image_points], len(image_points))
    if valid.sum() <= 1:
        return [None, None, None]
    ctx = default_context()
    _sync_cameras(ctx, camera_poses)
    xyz, ok = ctx.triangulate_batch(pts, valid, compact_k=True)
    return xyz[0]


def triangulate_points(image_points, camera_poses):
    """reference :87-99.  image_points shape = [obj points, camera_count, 2]; groups holding a [None, None] are
    skipped, so the result can have fewer rows than the input."""
    read_camera_params()
    full = [g for g in image_points if not any(_is_none(p) for p in g)]
    if len(full) == 0:
        return np.array([])
    n_cam = len(full[0])
    pts, valid = _pack_groups(full, n_cam)
    if n_cam <= 1:
        return np.array([[None, None, None]] * len(full))
    ctx = default_context()
    _sync_cameras(ctx, camera_poses)
    xyz, ok = ctx.triangulate_batch(pts, valid, compact_k=True)
    return xyz


def calculate_reprojection_errors(image_points, object_points, camera_poses):
    """reference :102-110.  Groups and object points are paired positionally (zip), as in the reference."""
    read_camera_params()
    pairs = list(zip(image_points, object_points))
    if len(pairs) == 0:
        return np.array([])
    n_cam = len(pairs[0][0])
    pts, valid = _pack_groups([g for g, _ in pairs], n_cam)
    xyz = np.array([np.asarray(o, float) for _, o in pairs]).reshape(-1, 3)
    ctx = default_context()
    _sync_cameras(ctx, camera_poses)
    mse, ok = ctx.reproject_batch(pts, valid, xyz, compact_k=True)
    return mse[ok != 0]


def calculate_reprojection_error(image_points, object_point, camera_poses):
    """reference :113-143.  image points shape (cam_count,2), object_point shape (3); None when fewer than two
    cameras see the point."""
    e = calculate_reprojection_errors([image_points], [object_point], camera_poses)
    return None if len(e) == 0 else e[0]


def _rotvec_to_matrix(v):
    """scipy Rotation.from_rotvec(v).as_matrix() (quaternion route)"""
    v = np.asarray(v, float)
    angle = np.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
    if angle <= 1e-3:
        a2 = angle * angle
        scale = 0.5 - a2 / 48 + a2 * a2 / 3840
    else:
        scale = np.sin(angle / 2) / angle
    x, y, z = scale * v
    w = np.cos(angle / 2)
    return np.array([[x * x - y * y - z * z + w * w, 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), -x * x + y * y - z * z + w * w, 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), -x * x - y * y + z * z + w * w]])


def params_to_camera_poses(params, num_cameras=2):
    """reference :145-156"""
    camera_poses = [{"R": np.eye(3), "t": np.array([0, 0, 0], dtype=np.float32)}]
    for i in range(0, num_cameras - 1):
        camera_poses.append({"R": _rotvec_to_matrix(params[i * 6: i * 6 + 3]), "t": params[i * 6 + 3: i * 6 + 6]})
    return camera_poses


def bundle_adjustment(image_points, camera_poses):
    """reference :158-176.  The residual vector (triangulate every point, reproject, per-point MSE as float32) is
    evaluated on the GPU; the trust-region driver is SciPy's, as in the reference.  Two cameras are assumed inside
    the residual, exactly as the reference hard-codes (:162,174).  The image points are uploaded once and stay resident
    (engine.BAProblem): an evaluation is one launch -- rotvec -> R, triangulation, reprojection and the float32 cast all
    on the device -- and one D2H of N floats."""
    from scipy import optimize
    from scipy.spatial.transform import Rotation

    read_camera_params()
    groups = [list(g) for g in image_points]
    problem = None
    if len(groups) and all(len(g) == 2 for g in groups):
        pts, valid = _pack_groups(groups, 2)
        ctx = default_context()
        _sync_cameras(ctx, params_to_camera_poses(np.zeros(6), 2))  # K and dist of cameras 0 and 1 (poses come from the parameters)
        problem = ctx.ba_problem(pts, valid)

    def residual_function(params):
        if problem is not None:
            return problem.residuals(np.asarray(params, float)[:6])
        poses = params_to_camera_poses(params, 2)  # groups of another width: the reference's own call chain
        object_points = triangulate_points(image_points, poses)
        errors = calculate_reprojection_errors(image_points, object_points, poses)
        return errors.astype(np.float32)

    init_params = np.array([])
    for camera_pose in camera_poses[1:]:
        init_params = np.concatenate([init_params, Rotation.from_matrix(camera_pose["R"]).as_rotvec(),
                                      np.asarray(camera_pose["t"]).flatten()])
    result = optimize.least_squares(residual_function, init_params, verbose=2, loss="linear", method='trf',
                                    ftol=1E-5, xtol=1E-15)
    return params_to_camera_poses(result.x)


def find_point_correspondance_and_object_points(image_points, camera_poses, obj_count=0, debug=False):
    """reference :178-280.  image_points shape = [camera_count, obj points, 2] (lists or arrays, the
    [[None, None]] sentinel of _find_dot allowed).  Returns (object points [<= obj_count+1, 3] sorted by mean
    reprojection error, image points of the surviving roots [roots, camera_count, 2]); both have shape (0,) when
    nothing triangulates.  The caller's lists lose their first [None, None] entry, as in the reference (:184-188)."""
    read_camera_params()
    for image_points_i in image_points:
        try:
            image_points_i.remove([None, None])
        except Exception:
            pass
    read_fundamental_matrix()
    n_cam = len(camera_poses)
    lists = [[p for p in cam if not _is_none(p)] for cam in image_points[:n_cam]]
    arrays = [np.asarray(l) for l in lists]
    is_int = all(a.size == 0 or a.dtype.kind in "iu" for a in arrays) and any(a.size for a in arrays)
    if is_int and any(a.size and np.abs(a).max() >= 2 ** 31 for a in arrays):
        is_int = False
    P = max(1, max(len(l) for l in lists))
    if P > 255:
        raise ValueError("more than 255 image points in one camera")
    pts = np.zeros((1, n_cam, P, 2), np.int32 if is_int else np.float64)
    counts = np.zeros((1, n_cam), np.int32)
    for c, a in enumerate(arrays):
        counts[0, c] = len(lists[c])
        if len(lists[c]):
            pts[0, c, : len(lists[c])] = a.reshape(-1, 2)
    if counts[0, 0] == 0 or n_cam < 2:
        return np.array([]), np.array([])
    ctx = default_context()
    _sync_cameras(ctx, camera_poses)
    ctx.set_fundamentals(np.asarray(Fs, float)[: n_cam - 1])
    out = ctx.correspond(torch.from_numpy(pts).to(ctx.device), torch.from_numpy(counts).to(ctx.device))
    n = int(out["n"].cpu()[0])
    if n < 0:
        raise RuntimeError(f"correspondence kernel capacity exceeded (status {n}); see MOCAP_CORR_E_GROUPS")
    if debug:
        print("Roots: ", out["root"].cpu().numpy()[0, :n].tolist())
    if n == 0:
        return np.array([]), np.array([])
    xyz = out["xyz"].cpu().numpy()[0, :n]
    grp = out["grp"].cpu().numpy()[0, :n]
    sorted_indices = out["order"].cpu().numpy()[0, :n]
    if not obj_count > len(sorted_indices):
        sorted_errors = sorted_indices[:obj_count + 1]
    else:
        sorted_errors = sorted_indices
    selected_object_points = xyz[sorted_errors]
    image_points_all = grp.astype(np.int64) if is_int else grp
    return np.array(selected_object_points), np.array(image_points_all)
