"""Parity of the HIP geometry path and of the drop-in lib.* surface with the reference-generated fixtures
(tests/golden, made by oracle/gen_golden.py from the reference's own lib/Helpers.py) and with the CPU oracle."""
import glob
import os

import numpy as np
import pytest

import oracle
from mocapv2_amd.synth import MILD_DIST, Scene

pytestmark = pytest.mark.gpu

TOL_XYZ = 1e-7  # world units = 1e-4 mm (BASELINE.json)
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def params_from(K, dist):
    return np.array([{"intrinsic_matrix": K[i].tolist(), "distortion_coef": dist[i].tolist()} for i in range(len(K))])


def poses_from(R, t):
    return [{"R": R[i], "t": t[i]} for i in range(len(R))]


@pytest.fixture()
def helpers():
    import torch
    assert torch.cuda.is_available()
    import mocapv2_amd.lib.Helpers as H
    H.camera_params = None
    H.Fs = []
    return H


def test_k1_triangulate_points(helpers):
    H = helpers
    g = load("k1_bundled")
    H.camera_params = params_from(g["K"], g["dist"])
    out = H.triangulate_points(g["image_points"], poses_from(g["R"], g["t"]))
    assert out.shape == (54, 3)
    assert np.abs(out - g["objects_json"]).max() < TOL_XYZ   # the reference's own bundled known answer
    assert np.abs(out - g["objects_ref"]).max() < 1e-9
    e = H.calculate_reprojection_errors(g["image_points"], out, poses_from(g["R"], g["t"]))
    assert np.allclose(e, g["reproj_mse"], rtol=1e-9, atol=1e-9)
    one = H.calculate_reprojection_error(g["image_points"][3], out[3], poses_from(g["R"], g["t"]))
    assert abs(one - g["reproj_mse"][3]) < 1e-9


def test_triangulate_point_none_handling(helpers):
    H = helpers
    g = load("tri_none")
    H.camera_params = params_from(g["K"], g["dist"])
    poses = poses_from(g["R"], g["t"])
    for grp, exp in zip(g["groups"], g["out"]):
        pts = [[None, None] if np.isnan(p[0]) else [int(p[0]), int(p[1])] for p in grp]
        got = H.triangulate_point(pts, poses)
        if np.isnan(exp[0]):
            assert list(got) == [None, None, None]
        else:
            assert np.abs(np.asarray(got, float) - exp).max() < TOL_XYZ
    # groups with a None are skipped by triangulate_points
    groups = [[[None, None] if np.isnan(p[0]) else [int(p[0]), int(p[1])] for p in grp] for grp in g["groups"]]
    out = H.triangulate_points(groups, poses)
    assert out.shape == (1, 3) and np.abs(out[0] - g["out"][0]).max() < TOL_XYZ
    assert H.triangulate_points(groups[1:], poses).shape == (0,)


def lists_from_fixture(g):
    lists = []
    for c in range(g["pts"].shape[0]):
        n = int(g["counts"][c])
        pts = g["pts"][c, :n]
        l = pts.tolist() if not bool(g["out_img_is_int"]) and n else [[int(x), int(y)] for x, y in pts]
        if bool(g["had_sentinel"][c]):
            l = [[None, None]] + l
        lists.append(l)
    return lists


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "corr_*.npz"))), ids=lambda p: os.path.basename(p)[:-4])
def test_correspondence_drop_in(helpers, path):
    H = helpers
    g = np.load(path)
    H.camera_params = params_from(g["K"], g["dist"])
    H.Fs = [F.tolist() for F in g["F"]]
    lists = lists_from_fixture(g)
    obj, img = H.find_point_correspondance_and_object_points(lists, poses_from(g["R"], g["t"]), int(g["obj_count"]))
    assert [len(l) for l in lists] == g["mutated_counts"].tolist()  # the sentinel was removed from the caller's lists
    assert obj.shape == tuple(g["out_obj_shape"]) and img.shape == tuple(g["out_img_shape"])
    if obj.shape == (0,):
        return
    assert (img.dtype.kind in "iu") == bool(g["out_img_is_int"])
    assert np.array_equal(img, g["out_img"])             # marker indices: bit-exact
    assert np.abs(obj - g["out_obj"]).max() < TOL_XYZ     # 3-D points: within 1e-4 mm


def test_ba_residual_vector(helpers):
    H = helpers
    g = load("ba_residuals")
    H.camera_params = params_from(g["K"], g["dist"])
    ip = g["image_points"]
    for x, r in zip(g["params"], g["residuals"]):
        poses = H.params_to_camera_poses(x, 2)
        obj = H.triangulate_points(ip, poses)
        e = H.calculate_reprojection_errors(ip, obj, poses).astype(np.float32)
        assert e.shape == r.shape and np.allclose(e, r, rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("C", [2, 3, 6, 8, 16])
def test_triangulation_against_scipys_svd_on_random_groups(C):
    """Evidence for the DLT that does not share a line with the kernel or the C oracle (whose Jacobi eigen-solve is the
    same code): 1500 random groups per camera count through mocap_triangulate_batch against the reference's own route
    (lib/Helpers.py:58-78: P = K [R|t], rows y P[2] - P[1] and P[0] - x P[2], B = A^T A, scipy.linalg.svd(B), Vh[3,:3] / Vh[3,3])
    evaluated with SciPy here, bar 1e-7 world units (1e-4 mm); integer pixel observations with noise, as the blob stage
    delivers them."""
    import scipy.linalg
    from mocapv2_amd.engine import MocapContext
    sc = Scene(C, 1920, 1080, dist=MILD_DIST, radius=3.0 if C <= 8 else 4.0)
    K = np.stack([sc.K] * C)
    R, t = np.stack([p["R"] for p in sc.poses]), np.stack([p["t"] for p in sc.poses])
    rng = np.random.default_rng(100 + C)
    N = 1500
    X = rng.uniform(-0.6, 0.6, (N, 3))
    pts = np.zeros((N, C, 2))
    for c in range(C):
        pc = X @ R[c].T + t[c]
        pts[:, c, 0] = K[c, 0, 0] * pc[:, 0] / pc[:, 2] + K[c, 0, 2]
        pts[:, c, 1] = K[c, 1, 1] * pc[:, 1] / pc[:, 2] + K[c, 1, 2]
    pts = np.floor(pts + rng.normal(0, 0.7, pts.shape))
    ctx = MocapContext(1, 1)
    ctx.set_cameras(K, np.zeros((C, 5)), R, t)
    got, ok = ctx.triangulate_batch(pts, np.ones((N, C), np.uint8), compact_k=True)
    assert ok.all()
    P = np.stack([K[c] @ np.c_[R[c], t[c]] for c in range(C)])
    worst = 0.0
    for n in range(N):
        A = np.concatenate([np.stack([pts[n, c, 1] * P[c][2] - P[c][1], P[c][0] - pts[n, c, 0] * P[c][2]]) for c in range(C)])
        _, _, Vh = scipy.linalg.svd(A.T @ A)
        worst = max(worst, np.abs(got[n] - Vh[3, :3] / Vh[3, 3]).max())
    assert worst < TOL_XYZ, worst


def test_resident_ba_residuals_match_the_reference_fixture_and_the_oracle():
    """mocap_ba_residuals (image points resident, rotvec -> R / triangulation / reprojection / float32 cast in one launch):
    the reference-generated residual vectors of tests/golden/ba_residuals.npz (lib/Helpers.py:161-167 run by
    oracle/gen_golden.py), one vector at a time and all of them as one batch; then 16 cameras x 64 points with [None, None]
    holes -- groups with a hole are not triangulated (:93) and the zip pairs what is left positionally (:104) -- against
    the oracle.  Tolerance: the residuals are float32 (rtol 2e-5: one float32 ulp of the MSE after the FP64 chain)."""
    from mocapv2_amd.engine import MocapContext
    g = load("ba_residuals")
    ctx = MocapContext(1, 1)
    ctx.set_cameras(g["K"], g["dist"], np.stack([np.eye(3)] * 2), np.zeros((2, 3)))
    prob = ctx.ba_problem(g["image_points"])
    for x, r in zip(g["params"], g["residuals"]):
        e = prob.residuals(x)
        assert e.dtype == np.float32 and e.shape == r.shape and np.allclose(e, r, rtol=2e-5, atol=1e-6)
    batch = prob.residuals(np.stack(g["params"]))
    for e, r in zip(batch, g["residuals"]):
        assert np.allclose(e, r, rtol=2e-5, atol=1e-6)
    # 16 cameras, 64 points, holes
    C, M = 16, 64
    sc = Scene(C, 3840, 2160, dist=MILD_DIST, radius=4.0)
    K, dist = np.stack([sc.K] * C), np.stack([sc.dist] * C)
    R, t = np.stack([p["R"] for p in sc.poses]), np.stack([p["t"] for p in sc.poses])
    rng = np.random.default_rng(8)
    cents = sc.centroids(sc.markers(rng, M, extent=1.2), rng, jitter=0.3)
    pts = np.stack([np.stack([cents[c][m] for c in range(C)]) for m in range(M)]).astype(float)
    from scipy.spatial.transform import Rotation
    base = []
    for c in range(1, C):
        Rrel = R[c] @ R[0].T
        base += list(Rotation.from_matrix(Rrel).as_rotvec()) + list(t[c] - Rrel @ t[0])
    sets = np.array(base) + rng.normal(0, 1e-3, (5, 6 * (C - 1)))
    ctx.set_cameras(K, dist, R, t)
    for holes in (0, 9):
        valid = np.ones((M, C), np.uint8)
        for _ in range(holes):
            valid[rng.integers(M), rng.integers(C)] = 0
        valid[3, :] = 0 if holes else 1      # a group nobody sees
        valid[7, 1:] = 0 if holes else 1     # a group seen by one camera only
        prob = ctx.ba_problem(pts, valid)
        got = prob.residuals(sets)
        for x, e in zip(sets, got):
            exp = oracle.ba_residuals(x, C, pts, valid, K, dist)
            assert e.shape == exp.shape and len(exp) > 40 and np.allclose(e, exp, rtol=2e-5, atol=1e-6), holes


@pytest.mark.parametrize("C,M,T", [(2, 5, 4), (6, 8, 16), (6, 32, 8), (8, 16, 4)])
def test_batched_correspond_matches_oracle(C, M, T):
    import torch
    from mocapv2_amd.engine import MocapContext
    sc = Scene(C, dist=MILD_DIST)
    K = np.stack([sc.K] * C)
    dist = np.stack([sc.dist] * C)
    R = np.stack([p["R"] for p in sc.poses])
    t = np.stack([p["t"] for p in sc.poses])
    F = np.stack(sc.Fs)
    P = M + 4
    pts = np.zeros((T, C, P, 2), np.int32)
    cnt = np.zeros((T, C), np.int32)
    for s in range(T):
        rng = np.random.default_rng(900 + s)
        cents = sc.centroids(sc.markers(rng, M), rng, jitter=0.6)
        for c in range(C):
            l = cents[c][rng.permutation(M)]
            if s % 3 == 1 and c == C - 1:
                l = l[: M - 2]  # some markers unseen by the last camera
            extra = rng.integers(0, 1000, (rng.integers(0, 4), 2))
            l = np.concatenate([l, extra])
            cnt[s, c] = len(l)
            pts[s, c, : len(l)] = l
    ctx = MocapContext(1, 1)
    ctx.set_cameras(K, dist, R, t)
    ctx.set_fundamentals(F)
    out = ctx.correspond(torch.from_numpy(pts).cuda(), torch.from_numpy(cnt).cuda())
    n = out["n"].cpu().numpy()
    for s in range(T):
        ref = oracle.correspond(pts[s].astype(float), cnt[s], K, dist, R, t, F)
        assert n[s] == len(ref["root"])
        k = n[s]
        assert np.array_equal(out["root"][s, :k].cpu().numpy(), ref["root"])
        assert np.array_equal(out["grp"][s, :k].cpu().numpy(), ref["groups"])
        assert np.array_equal(out["order"][s, :k].cpu().numpy(), ref["order"])
        assert np.abs(out["xyz"][s, :k].cpu().numpy() - ref["xyz"]).max() < TOL_XYZ
        assert np.allclose(out["err"][s, :k].cpu().numpy(), ref["err"], rtol=1e-9, atol=1e-12)
    assert n.sum() > 0


def test_cuda_operations_drop_in():
    from mocapv2_amd.lib.CudaOperations import fast_cuda_blur, fast_cuda_demosaic
    rng = np.random.default_rng(0)
    for H_, W_ in [(1, 1), (17, 23), (120, 200)]:
        img = rng.integers(0, 256, (H_, W_), dtype=np.uint8)
        out = fast_cuda_blur(img, 5)
        assert out.dtype == np.uint8 and np.array_equal(out, oracle.box_blur(img, 5))
        assert np.array_equal(fast_cuda_blur(img, 3), oracle.box_blur(img, 3))
        assert np.array_equal(fast_cuda_demosaic(img), oracle.demosaic(img))
    with pytest.raises(AssertionError):
        fast_cuda_blur(np.zeros((4, 4, 3), np.uint8))


def test_image_operations_drop_in():
    import mocapv2_amd.lib.ImageOperations as IO
    sc = Scene(2, width=640, height=360, dist=MILD_DIST)
    rng = np.random.default_rng(4)
    img = sc.render(rng, sc.markers(rng, 5, extent=1.0), 0, radius_range=(16, 20), salt=0.001)
    IO.camera_params = [{"intrinsic_matrix": sc.K.tolist(), "distortion_coef": sc.dist.tolist()}]
    assert np.array_equal(IO.image_filter_gpu(img), oracle.image_filter(img, 0))
    assert np.array_equal(IO.image_filter_cpu(img), oracle.image_filter(img, 1))
    out, pts = IO._find_dot(img)
    assert pts == oracle.find_dot(img, sc.K, sc.dist) and len(pts) == 5
    assert np.array_equal(out, oracle.undistort(img, sc.K, sc.dist))
    filt, pts2 = IO._find_dot(img, return_filtered=True)
    assert pts2 == pts and np.array_equal(filt, oracle.image_filter(oracle.undistort(img, sc.K, sc.dist), 0))
    _, none = IO._find_dot(np.zeros((360, 640), np.uint8))
    assert none == [[None, None]]


def test_camera_threads_as_in_the_reference_tracker(helpers):
    """The reference's process layout (RealtimeTracking_FLIR.py:307-312): one thread per camera loops over `_find_dot`
    and queues its image points, one consumer thread pairs them and calls
    `find_point_correspondance_and_object_points`.  Same layout here, all threads running at once (ctypes drops the
    GIL inside the library): every thread works on its own context and the results equal the sequential ones."""
    import queue
    import threading
    import mocapv2_amd.lib.ImageOperations as IO
    C, T = 4, 6
    sc = Scene(C, width=640, height=360, dist=MILD_DIST)
    IO.camera_params = [{"intrinsic_matrix": sc.K.tolist(), "distortion_coef": sc.dist.tolist()}]
    helpers.camera_params = np.array(sc.camera_params)
    helpers.Fs = [F.tolist() for F in sc.Fs]
    frames = []
    for t_ in range(T):
        rng = np.random.default_rng(300 + t_)
        mk = sc.markers(rng, 4, extent=0.8)
        frames.append([sc.render(rng, mk, c, radius_range=(16, 19), salt=0.001) for c in range(C)])
    sequential = [[IO._find_dot(frames[t_][c])[1] for c in range(C)] for t_ in range(T)]
    seq_obj = [helpers.find_point_correspondance_and_object_points([list(l) for l in sequential[t_]], sc.poses, 4)
               for t_ in range(T)]

    queues = [queue.Queue() for _ in range(C)]
    errors = []
    start = threading.Barrier(C + 1)

    def track_points(c):
        try:
            start.wait()
            for t_ in range(T):
                queues[c].put(IO._find_dot(frames[t_][c])[1])
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            queues[c].put(None)

    results = []

    def track():
        try:
            start.wait()
            for t_ in range(T):
                pts = [queues[c].get(timeout=120) for c in range(C)]
                results.append((pts, helpers.find_point_correspondance_and_object_points([list(l) for l in pts], sc.poses, 4)))
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=track_points, args=(c,)) for c in range(C)] + [threading.Thread(target=track)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(300)
    assert not errors, errors
    assert len(results) == T
    for t_ in range(T):
        assert results[t_][0] == sequential[t_]
        assert np.array_equal(results[t_][1][0], seq_obj[t_][0]) and np.array_equal(results[t_][1][1], seq_obj[t_][1])
        assert len(seq_obj[t_][0]) > 0


def test_sharded_pipeline_equals_single_rank():
    """The N-rank layout (camera-major blocks, per-segment launches with slot_base, strided reads of the gathered
    records, time-sliced triangulation) gives exactly the single-rank results.  Both 'ranks' run on this one GPU and
    the all-gather is replaced by the concatenation it produces."""
    import torch
    from mocapv2_amd.pipeline import BatchTracker, scene_arrays
    C, T, W, H = 3, 4, 640, 360
    sc = Scene(C, W, H, dist=MILD_DIST)
    arrays = scene_arrays(sc)
    world = 2

    def frame(c, t):
        rng = np.random.default_rng(50 + t)
        mk = sc.markers(rng, 4, extent=0.8)
        return sc.render(np.random.default_rng(1000 * t + c), mk, c, radius_range=(16, 20))

    # reference: one rank, all T*world time steps
    ref = BatchTracker(*arrays, W, H, T * world)
    fr = np.stack([frame(c, t) for (c, t) in ref.local_image_list()])
    ref_out = {k: v.cpu().numpy() for k, v in ref.step(torch.from_numpy(fr).cuda()).items()}
    assert (ref_out["n"] > 0).any()
    # two ranks
    trackers = [BatchTracker(*arrays, W, H, T, world=world, rank=r) for r in range(world)]
    recs = []
    for tr in trackers:
        fr_r = np.stack([frame(c, t) for (c, t) in tr.local_image_list()])
        recs.append(tr.extract(torch.from_numpy(fr_r).cuda()).clone())
    gathered = torch.cat(recs, dim=0)  # what all_gather_into_tensor leaves on every rank
    for r, tr in enumerate(trackers):
        out = {k: v.cpu().numpy() for k, v in tr.triangulate(gathered).items()}
        sl = slice(r * T, (r + 1) * T)
        assert np.array_equal(out["n"], ref_out["n"][sl])
        for s in range(T):
            k = out["n"][s]
            assert np.array_equal(out["grp"][s, :k], ref_out["grp"][sl][s, :k])
            assert np.array_equal(out["xyz"][s, :k], ref_out["xyz"][sl][s, :k])  # same kernel, same inputs: bit-equal
            assert np.array_equal(out["order"][s, :k], ref_out["order"][sl][s, :k])


@pytest.mark.parametrize("collective", ["rccl", "torch"])
def test_one_rank_collective_through_rccl_equals_no_collective(collective):
    """As much RCCL as one GPU allows: BatchTracker with three batches in flight and the exchange step forced on with
    world = 1 -- through mocap_allgather_centroids (ncclCommInitRank + ncclAllGather behind the C-ABI, one communicator
    per batch in flight, issued on the batch's own HIP stream) and through torch.distributed's nccl backend
    (all_gather_into_tensor) -- gives bit for bit the results of the path without a collective."""
    import torch
    import torch.distributed as dist
    from mocapv2_amd.pipeline import BatchTracker, scene_arrays
    C, T, W, H = 3, 4, 640, 360
    sc = Scene(C, W, H, dist=MILD_DIST)
    arrays = scene_arrays(sc)
    batches = [torch.from_numpy(sc.render_batch(seed=70 + b, n_steps=T, n_markers=4, radius_range=(16, 20)).reshape(T * C, H, W)).cuda()
               for b in range(5)]
    ref = BatchTracker(*arrays, W, H, T)
    expected = []
    for fr in batches:
        out = ref.step(fr)
        torch.cuda.synchronize()
        expected.append({k: v.cpu().numpy().copy() for k, v in out.items()})
    own_pg = False
    if collective == "torch" and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        own_pg = True
    try:
        trk = BatchTracker(*arrays, W, H, T, depth=3, collective=collective, force_collective=True)
        assert trk.collective == collective
        outs = []
        for fr in batches:
            out = trk.step(fr)
            n = trk.finish(out)  # the lane's buffers are reused three batches later: copy the results out now
            outs.append({k: v.cpu().numpy().copy() for k, v in out.items()})
            assert (n >= 0).all()
        if collective == "rccl":
            for lane in trk.lanes:  # the gathered copy is what correspondence read
                assert torch.equal(lane.gathered, lane.records)
    finally:
        if own_pg:
            dist.destroy_process_group()
    assert any((e["n"] > 0).any() for e in expected)
    for e, o in zip(expected, outs):
        assert np.array_equal(e["n"], o["n"])
        for s in range(T):
            k = e["n"][s]
            assert np.array_equal(e["xyz"][s, :k], o["xyz"][s, :k]) and np.array_equal(e["grp"][s, :k], o["grp"][s, :k])
            assert np.array_equal(e["order"][s, :k], o["order"][s, :k])


def test_replay_tracker_matches_per_frame_drop_in(helpers):
    """The batched headless tracker gives, per time step, what the reference's loops give for the same frames:
    _find_dot per camera, find_point_correspondance_and_object_points(..., 4), the msgpack message (with the
    previous point repeated when nothing was found)."""
    import mocapv2_amd.lib.ImageOperations as IO
    from mocapv2_amd.pipeline import scene_arrays
    from mocapv2_amd.replay import ReplayTracker, tracker_message
    H = helpers
    C, T, W, Hh = 2, 5, 640, 360
    sc = Scene(C, W, Hh, dist=MILD_DIST)
    K, dist, R, t, F = scene_arrays(sc)
    frames = sc.render_batch(seed=21, n_steps=T, n_markers=3, radius_range=(16, 20))
    frames[2] = 10  # a time step without any marker: the previous message must be repeated
    params = [{"intrinsic_matrix": sc.K.tolist(), "distortion_coef": sc.dist.tolist()} for _ in range(C)]
    IO.camera_params = params
    H.camera_params = np.array(params)
    H.Fs = [f.tolist() for f in F]
    poses = poses_from(R, t)
    point = [0, 0, 0, 0, 0, 0, 0, 0]
    got = list(ReplayTracker(K, dist, R, t, F, W, Hh, batch=4).run(frames))  # 5 steps in batches of 4: one padded batch
    assert len(got) == T
    seen = 0
    for s in range(T):
        lists = [IO._find_dot(frames[s, c])[1] for c in range(C)]
        obj, img = H.find_point_correspondance_and_object_points(lists, poses, 4)
        if len(obj) > 0:
            point = [0, 0, 0, 0] + list(obj[0])
            seen += 1
        assert got[s]["object_points"].shape == obj.shape and np.array_equal(got[s]["object_points"], obj), s
        assert got[s]["image_points"].shape == img.shape and np.array_equal(got[s]["image_points"], img), s
        assert got[s]["message"] == tracker_message(point), s
    assert seen >= 3 and len(got[2]["object_points"]) == 0 and got[2]["message"] == got[1]["message"]


def test_host_fed_batches_equal_resident_ones():
    """BatchTracker.step accepts a batch in pinned host memory (uploaded on the batch's own stream, three batches in
    flight) and gives the results of the same batch resident on the GPU."""
    import torch
    from mocapv2_amd.pipeline import BatchTracker, scene_arrays
    C, T, W, H = 2, 3, 640, 360
    sc = Scene(C, W, H, dist=MILD_DIST)
    arrays = scene_arrays(sc)
    batches = [sc.render_batch(seed=30 + b, n_steps=T, n_markers=4, radius_range=(16, 20)).reshape(T * C, H, W) for b in range(4)]
    ref = BatchTracker(*arrays, W, H, T)
    expected = []
    for fr in batches:
        out = ref.step(torch.from_numpy(fr).cuda())
        torch.cuda.synchronize()
        expected.append({k: v.cpu().numpy().copy() for k, v in out.items()})
    trk = BatchTracker(*arrays, W, H, T, depth=3)
    outs = []
    for fr in batches:
        out = trk.step(torch.from_numpy(fr).pin_memory())
        trk.synchronize()  # the lane's buffers are reused three batches later: copy the results out now
        outs.append({k: v.cpu().numpy().copy() for k, v in out.items()})
    for e, o in zip(expected, outs):
        assert np.array_equal(e["n"], o["n"]) and (e["n"] > 0).any()
        for s in range(T):
            k = e["n"][s]
            assert np.array_equal(e["xyz"][s, :k], o["xyz"][s, :k]) and np.array_equal(e["grp"][s, :k], o["grp"][s, :k])


# ---- calibration-time batch paths (mocapv2_amd/calibrate.py) -----------------------------------------
def test_cheirality_vote_matches_reference_fixture(helpers):
    """The four candidates of the bundled capture in one launch: object points, vote counts and winner equal what the
    reference's triangulate_points + the vote of CalculateCameraPoses.py:199-231 produced (calib_cheirality.npz)."""
    from mocapv2_amd import calibrate as cal
    g = load("calib_cheirality")
    params = params_from(g["K"], g["dist"])
    R1, R2, t = cal.decompose_essential(g["K"][1].T @ g["F"] @ g["K"][0])
    base = {"R": np.eye(3), "t": np.zeros((3, 1), np.float32)}
    ip = g["image_points"]
    out = cal.select_relative_pose(ip[:, 0], ip[:, 1], base, R1, R2, t, params)
    # map this decomposition's candidate order onto the fixture's (the SVD's sign freedom may permute it)
    cand_R, cand_t = [R1, R1, R2, R2], [t, -t, t, -t]
    for i in range(4):
        j = [k for k in range(4) if np.abs(cand_R[i] - g["cand_R"][k]).max() < 1e-12
             and np.abs(cand_t[i].ravel() - g["cand_t"][k]).max() < 1e-12]
        assert len(j) == 1
        assert np.abs(out["object_points"][i] - g["objects"][j[0]]).max() < TOL_XYZ
        assert out["counts"][i] == int(g["counts"][j[0]])
    assert np.abs(out["pose"]["R"] - g["chosen_R"]).max() < 1e-12
    assert np.abs(out["pose"]["t"].ravel() - g["chosen_t"]).max() < 1e-12
    # brute force through the drop-in, one call per candidate, as the reference does it
    helpers.camera_params = params
    for i in range(4):
        o = helpers.triangulate_points(ip, [base, {"R": cand_R[i], "t": cand_t[i]}])
        assert np.array_equal(o, out["object_points"][i])


def test_extrinsics_chain_equals_the_reference_loop(helpers):
    """Three synthetic cameras, F(i -> i+1) from the true poses, sub-pixel image points.  The batched chain equals the
    reference's loop (four `triangulate_points` calls per link through the drop-in, the vote of :214-224) pose for
    pose; rotations are the true relative rotations, translations the unit baseline up to the sign the reference's
    vote -- which looks at (R^T X).z, not at the depth in the second camera -- leaves open."""
    from mocapv2_amd import calibrate as cal
    from mocapv2_amd.synth import ZERO_DIST, fundamental_from_poses, project
    sc = Scene(3, dist=ZERO_DIST)
    rng = np.random.default_rng(21)
    X = sc.markers(rng, 24)
    K = np.asarray(sc.camera_params[0]["intrinsic_matrix"], float)
    pts = [project(X, sc.poses[c], K, ZERO_DIST) for c in range(3)]
    Fs = [fundamental_from_poses(sc.poses[i], sc.poses[i + 1], K, K) for i in range(2)]
    poses = cal.extrinsics_from_fundamentals(pts, Fs, sc.camera_params)
    assert len(poses) == 3

    helpers.camera_params = sc.camera_params
    want = [{"R": np.eye(3), "t": np.zeros((3, 1))}]
    for i in range(2):
        R1, R2, t = cal.decompose_essential(K.T @ Fs[i] @ K)
        best, most = None, 0
        for R, tt in zip([R1, R1, R2, R2], [t, -t, t, -t]):
            p = np.transpose([pts[i], pts[i + 1]], [1, 0, 2])
            o = helpers.triangulate_points(p, [want[-1], {"R": R, "t": tt}])
            oc = np.array([R.T @ q for q in o])
            n = np.sum(o[:, 2] > 0) + np.sum(oc[:, 2] > 0)
            if n > most:
                best, most = (R, tt), n
        want.append({"R": best[0] @ want[-1]["R"], "t": want[-1]["t"] + want[-1]["R"] @ best[1]})
    for got, w in zip(poses, want):
        assert np.array_equal(got["R"], w["R"]) and np.array_equal(got["t"], w["t"])

    R0, t0 = sc.poses[0]["R"], sc.poses[0]["t"].reshape(3)
    for i in (1, 2):
        assert np.abs(poses[i]["R"] - sc.poses[i]["R"] @ R0.T).max() < 1e-9
    t_rel = sc.poses[1]["t"].reshape(3) - sc.poses[1]["R"] @ R0.T @ t0
    assert min(np.abs(s * poses[1]["t"].ravel() - t_rel / np.linalg.norm(t_rel)).max() for s in (1, -1)) < 1e-9


def test_batched_jacobian_is_scipys_two_point_jacobian():
    """One triangulation + one reprojection launch over 7 parameter vectors give bit for bit the Jacobian SciPy's
    `least_squares(jac='2-point')` derives by calling the residual 6 more times."""
    from scipy.optimize._numdiff import approx_derivative
    from mocapv2_amd import calibrate as cal
    g = load("ba_residuals")
    params = params_from(g["K"], g["dist"])
    ip = g["image_points"]
    for x0, want in zip(g["params"], g["residuals"]):
        f0, J = cal.residual_and_jacobian(ip, x0, params)
        assert f0.dtype == np.float32
        assert np.allclose(f0, want, rtol=2e-6, atol=0)  # the reference's residual vector (float32)
        single = lambda x: cal.residuals_batched(ip, [x], params)[0]  # noqa: E731
        assert np.array_equal(single(x0), f0)
        assert np.array_equal(J, approx_derivative(single, x0, method="2-point", f0=f0))


def test_bundle_adjustment_reproduces_the_bundled_after_ba_extrinsics():
    """jsons/before_ba_extrinsics.json --BA--> jsons/after_ba_extrinsics.json is the reference's own known answer for
    the loop of lib/Helpers.py:158-176 (re-running the reference here lands on that file exactly)."""
    from mocapv2_amd import calibrate as cal
    g = load("calib_cheirality")
    k1 = load("k1_bundled")
    params = params_from(g["K"], g["dist"])
    before = [{"R": np.eye(3), "t": np.zeros(3)}, {"R": g["before_ba_R"], "t": g["before_ba_t"]}]
    runs = []
    for batched in (True, False):
        poses, result = cal.bundle_adjustment(g["image_points"], before, params, batched_jacobian=batched)
        runs.append((poses, result))
        assert result.status > 0
        assert np.abs(poses[1]["R"] - k1["R"][1]).max() < 1e-6
        assert np.abs(np.asarray(poses[1]["t"]) - k1["t"][1]).max() < 1e-6
    assert np.array_equal(runs[0][1].x, runs[1][1].x) and runs[0][1].nfev == runs[1][1].nfev
