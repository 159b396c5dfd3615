"""Worker of tests/test_gpu_two_ranks.py: rank `argv[1]` of a world of 2 whose ranks share cuda:0 (process group over gloo).
Builds BatchTracker(world=2, collective="rccl"): the library's staged set-up (pipeline.negotiate_rccl) runs against the real
RCCL -- two ranks on ONE device is a configuration RCCL refuses, so the collective mocap_comm_init fails on both ranks, the
ranks agree on that and fall back to torch.distributed together -- then one batch, compared with a single-rank tracker."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocapv2_amd.pipeline import BatchTracker, scene_arrays  # noqa: E402
from mocapv2_amd.synth import MILD_DIST, Scene  # noqa: E402


def main():
    rank, port = int(sys.argv[1]), int(sys.argv[2])
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    C, T, W, H, world = 4, 3, 640, 360, 2
    sc = Scene(C, W, H, dist=MILD_DIST)
    arrays = scene_arrays(sc)

    def frame(c, t):
        mk = sc.markers(np.random.default_rng(50 + t), 4, extent=0.8)
        return sc.render(np.random.default_rng(1000 * t + c), mk, c, radius_range=(16, 20))

    ref = BatchTracker(*arrays, W, H, T * world)
    fr = np.stack([frame(c, t) for (c, t) in ref.local_image_list()])
    ref_out = {k: v.cpu().numpy() for k, v in ref.step(torch.from_numpy(fr).cuda()).items()}
    trk = BatchTracker(*arrays, W, H, T, world=world, rank=rank, collective="rccl", depth=2)
    flags = [None, None]
    dist.all_gather_object(flags, trk.collective)
    assert flags[0] == flags[1], flags  # both ranks took the same road
    fr_r = torch.from_numpy(np.stack([frame(c, t) for (c, t) in trk.local_image_list()])).cuda()
    for _ in range(3):  # both lanes
        out = trk.step(fr_r)
    trk.synchronize()
    out = {k: v.cpu().numpy() for k, v in out.items()}
    sl = slice(rank * T, (rank + 1) * T)
    assert np.array_equal(out["n"], ref_out["n"][sl]) and (out["n"] > 0).any()
    for s in range(T):
        k = out["n"][s]
        assert np.array_equal(out["grp"][s, :k], ref_out["grp"][sl][s, :k])
        assert np.array_equal(out["xyz"][s, :k], ref_out["xyz"][sl][s, :k])
        assert np.array_equal(out["order"][s, :k], ref_out["order"][sl][s, :k])
    print(f"OK rank {rank} collective={trk.collective}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
