"""Multi-process (gloo, CPU) coverage of the N-GPU path: the camera-major shard plan and the single all-gather of
centroid records.  No compute runs here (there is no CPU fallback): ranks fabricate their records deterministically,
exchange them exactly as BatchTracker.step does, and check what every rank ends up holding."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mocapv2_amd.pipeline import allgather_records, negotiate_rccl, shard_plan


def test_shard_plan_covers_the_grid_exactly_once():
    for n_cams, T, world in [(6, 64, 1), (6, 64, 2), (6, 64, 4), (6, 64, 8), (8, 16, 8), (16, 5, 3), (1, 7, 4)]:
        t_total = T * world
        seen = np.zeros((n_cams, t_total), int)
        for r in range(world):
            segs = shard_plan(n_cams, T, world, r)
            assert sum(t1 - t0 for _, t0, t1 in segs) == n_cams * T  # weak scaling: equal work per rank
            flat = []
            for c, t0, t1 in segs:
                seen[c, t0:t1] += 1
                flat += [c * t_total + t for t in range(t0, t1)]
            assert flat == list(range(r * n_cams * T, (r + 1) * n_cams * T))  # contiguous camera-major block
        assert (seen == 1).all()


def fake_record(c, t, rec_ints):
    rng = np.random.default_rng(c * 100003 + t)
    rec = np.zeros(rec_ints, np.int32)
    n = int(rng.integers(0, 9))
    rec[0] = n
    rec[2:2 + 2 * n] = rng.integers(0, 1920, 2 * n)
    return rec


def _worker(rank, world, port, n_cams, T, rec_ints, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        segs = shard_plan(n_cams, T, world, rank)
        local = np.stack([fake_record(c, t, rec_ints) for c, t0, t1 in segs for t in range(t0, t1)])
        gathered = allgather_records(torch.from_numpy(local), world).numpy()
        t_total = T * world
        ok = gathered.shape == (n_cams * t_total, rec_ints)
        # every rank holds every camera's record for every time step, in camera-major order
        for c in range(n_cams):
            for t in range(0, t_total, 3):
                ok &= bool(np.array_equal(gathered[c * t_total + t], fake_record(c, t, rec_ints)))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_cams,T", [(2, 6, 8), (3, 4, 5)])
def test_allgather_of_centroid_records_gloo(world, n_cams, T):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_cams, T, 2 + 2 * 16, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(r, True) for r in range(world)]


SCENARIOS = ["ok", "prepare_fails_rank1", "uid_fails_rank0", "init_fails_rank1", "share_fails_rank0"]


def _negotiate_worker(rank, world, port, q):
    """Every scenario injects one failure on one rank into the set-up of the RCCL exchange (BatchTracker._setup_collective's
    negotiate_rccl) with stand-ins for the C-ABI calls; the stand-in for mocap_comm_init holds a real collective (as
    ncclCommInitRank does), so a rank that skipped it -- or entered it alone -- would hang the test."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        for sc in SCENARIOS:
            log = []

            def prepare():
                log.append("prepare")
                if sc == "prepare_fails_rank1" and rank == 1:
                    raise RuntimeError("librccl.so.1 not found (injected)")

            def unique_id():
                log.append("uid")
                if sc == "uid_fails_rank0":
                    raise RuntimeError("ncclGetUniqueId failed (injected)")
                return bytes(range(1, 129))

            def comm_init(uid):
                log.append("init")
                assert uid == bytes(range(1, 129))
                dist.barrier()  # ncclCommInitRank is collective: every rank must be here, or nobody
                if sc == "init_fails_rank1" and rank == 1:
                    raise RuntimeError("ncclCommInitRank failed (injected)")

            def share():
                log.append("share")
                if sc == "share_fails_rank0" and rank == 0:
                    raise RuntimeError("mocap_comm_share failed (injected)")

            ok = negotiate_rccl(world, rank, prepare=prepare, unique_id=unique_id, comm_init=comm_init, share=share,
                                destroy=lambda: log.append("destroy"), release=lambda: log.append("release"), dist=dist)
            q.put((rank, sc, ok, log))
    finally:
        dist.destroy_process_group()


def test_rccl_setup_agrees_on_every_rank_when_one_rank_fails():
    """ADVICE r2 / VERDICT r2 item 8b: whichever rank fails at whichever stage, every rank ends on the same transport, nobody
    waits in a collective a peer never enters, and a communicator that was created on one side is destroyed again."""
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_negotiate_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world * len(SCENARIOS)):
        rank, sc, ok, log = q.get(timeout=120)
        got[(sc, rank)] = (ok, log)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        assert got[("ok", r)] == (True, ["prepare"] + (["uid"] if r == 0 else []) + ["init", "share"])
        # a local failure is agreed on before anybody enters the collective init
        assert got[("prepare_fails_rank1", r)] == (False, ["prepare", "release"])
        assert got[("uid_fails_rank0", r)] == (False, ["prepare"] + (["uid"] if r == 0 else []) + ["release"])
        # init: every rank was in it; the rank that holds a communicator gives it back
        assert got[("init_fails_rank1", r)] == (False, ["prepare"] + (["uid"] if r == 0 else []) + ["init"] + (["destroy"] if r == 0 else []) + ["release"])
        assert got[("share_fails_rank0", r)] == (False, ["prepare"] + (["uid"] if r == 0 else []) + ["init", "share", "destroy", "release"])


def test_rccl_setup_single_rank_without_a_process_group():
    log = []
    ok = negotiate_rccl(1, 0, prepare=lambda: log.append("prepare"), unique_id=lambda: bytes(128) [:0] + b"\x01" * 128,
                        comm_init=lambda uid: log.append(("init", len(uid))), share=lambda: log.append("share"),
                        destroy=lambda: log.append("destroy"), release=lambda: log.append("release"))
    assert ok and log == ["prepare", ("init", 128), "share"]


def test_bench_counts_gpus_without_touching_hip(monkeypatch):
    """bench.py's launcher parent must not initialise HIP before it starts the rank processes."""
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    assert bench.visible_gpu_count() == 3
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    n = bench.visible_gpu_count()
    assert n is None or n >= 0
    import inspect
    src = inspect.getsource(bench.self_launch) + inspect.getsource(bench.visible_gpu_count)
    assert "import torch" not in src and "device_count" not in src


def test_single_rank_gathers_nothing():
    x = torch.arange(12, dtype=torch.int32).reshape(3, 4)
    assert allgather_records(x, 1) is x


def test_tracker_wire_format():
    """msgpack bytes of one tracker update as the reference sends them (RealtimeTracking_FLIR.py:171,183-187)."""
    import struct
    from mocapv2_amd.replay import tracker_message, unpack_tracker_message
    assert tracker_message([0] * 8) == b"\x81\xa8tracker1\x98" + b"\x00" * 8  # before the first detection: eight zeros
    p = [0, 0, 0, 0, np.float64(0.25), np.float64(-1.5), np.float64(3.0000001)]
    exp = b"\x81\xa8tracker1\x97" + b"\x00" * 4 + b"".join(b"\xcb" + struct.pack(">d", float(v)) for v in p[4:])
    assert tracker_message(p) == exp
    assert unpack_tracker_message(exp) == [0, 0, 0, 0, 0.25, -1.5, 3.0000001]


def test_bulk_messages_equal_msgpack_per_time_step():
    """replay.batch_messages builds the tracker messages of a whole batch with NumPy; every one of them must be byte for byte what
    msgpack.packb({"tracker1": [0, 0, 0, 0, x, y, z]}, use_bin_type=True) gives (RealtimeTracking_FLIR.py:186-187), time steps
    without a point repeat the message in force (:181-188), and before the first detection that is the eight-zeros list (:171)."""
    from mocapv2_amd.replay import batch_messages, tracker_message
    rng = np.random.default_rng(0)
    pts = rng.normal(0, 5, (64, 3))
    pts[3] = [np.nan, -0.0, np.inf]
    pts[7] = [1e-310, -1e308, 0.0]
    has = rng.random(64) < 0.6
    has[:2] = False
    carry = tracker_message([0] * 8)
    got = batch_messages(pts, has, carry)
    point = [0] * 8
    for i in range(64):
        if has[i]:
            point = [0, 0, 0, 0] + list(pts[i])
        assert got[i] == tracker_message(point), i
    assert batch_messages(np.zeros((0, 3)), np.zeros(0, bool), carry) == []
