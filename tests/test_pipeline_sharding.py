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

from mocapv2_amd.pipeline import allgather_records, shard_plan


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
