"""Batched tracker: the build's counterpart of the reference's track_points / track loops
(reference RealtimeTracking_FLIR.py:95-143,157-209), fed from frames resident in HBM instead of PySpin.

One process per GPU.  The grid of (camera, time step) images is cut camera-major into equal contiguous blocks,
one per rank (a rank holds whole cameras, or a run of time steps of one, never a mix it does not need); each rank
extracts the 2-D centroids of its block, ONE all-gather of fixed-size centroid records (RCCL over xGMI) gives every
rank all centroids, and each rank then runs correspondence + triangulation for its own slice of the time steps.
With one rank the block is the whole grid and no collective runs.
"""
import numpy as np
import torch

from .engine import GRAY_SHIFT, MocapContext


def shard_plan(n_cams, steps_per_rank, world, rank):
    """Segments (camera, t_begin, t_end) of the camera-major block of `rank`.

    The run covers T_total = steps_per_rank * world time steps (weak scaling: n_cams * steps_per_rank images per
    rank whatever the world size).  Image (c, t) has flat index c * T_total + t; rank r owns
    [r * per, (r + 1) * per) with per = n_cams * steps_per_rank."""
    t_total = steps_per_rank * world
    per = n_cams * steps_per_rank
    lo, hi = rank * per, (rank + 1) * per
    segs = []
    idx = lo
    while idx < hi:
        c, t = divmod(idx, t_total)
        t_end = min(t_total, t + (hi - idx))
        segs.append((c, t, t_end))
        idx += t_end - t
    return segs


def allgather_records(local, world, group=None, force=False):
    """The path's only collective, host-side form (torch.distributed): every rank contributes its [per, REC] centroid
    records, every rank receives [world * per, REC] in rank order = camera-major order.  `local` may live on the GPU
    (nccl = RCCL) or on the CPU (gloo, used by the multi-process tests).  The product's default on GPUs is the same
    all-gather behind the C-ABI (`MocapContext.allgather_centroids` = mocap_allgather_centroids, RCCL called
    directly on the batch's HIP stream); this form serves gloo rehearsals / CPU tests and `collective="torch"`.
    force: run the collective even with one rank (tests)."""
    if world == 1 and not force:
        return local
    import torch.distributed as dist
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal of the multi-rank path on a box without RCCL peers: gloo has no GPU all-gather, stage through the host
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather(list(host.chunk(world, dim=0)), local.cpu().contiguous(), group=group)
        out.copy_(host)
    elif local.is_cuda:
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    else:
        dist.all_gather(list(out.chunk(world, dim=0)), local.contiguous(), group=group)
    return out


def negotiate_rccl(world, rank, prepare, unique_id, comm_init, share, destroy, release, dist=None, group=None, flag_device="cpu"):
    """Set up the RCCL exchange on every rank or on none: True when all ranks hold a communicator, False when all must take
    the torch.distributed road instead.  A rank that fails never leaves its peers inside a collective it does not enter:
      1. `prepare()` -- everything that can fail locally (librccl loadable, buffers allocated) -- runs BEFORE any collective,
         and its outcome is agreed on (all-reduce MIN) before any rank goes on;
      2. rank 0's `unique_id()` travels in ONE broadcast that every rank enters whatever happened (zeros = failed), and the
         outcome is agreed on again;
      3. only then every rank calls `comm_init(uid)` (ncclCommInitRank: collective inside RCCL, all ranks are in it), and the
         outcome is agreed on a third time; ranks that did get a communicator `destroy()` it when a peer did not;
      4. `share()` (local) hands the communicator to the rank's other batches in flight.
    `dist`: torch.distributed (or a stand-in with broadcast / all_reduce / ReduceOp / get_global_rank) when world > 1, else None.
    The callables are injected so that the agreement logic is tested with gloo on CPUs (tests/test_pipeline_sharding.py)."""
    import sys
    from . import _abi

    def say(what, e):
        print(f"[mocapv2_amd] rank {rank}: {what} failed ({e}); exchanging through torch.distributed instead", file=sys.stderr)

    def agree(ok):
        if dist is None:
            return ok
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=flag_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        return bool(flag.item())

    try:
        prepare()
        ok = True
    except Exception as e:  # noqa: BLE001
        say("preparing the RCCL exchange", e)
        ok = False
    if not agree(ok):
        release()
        return False
    uid = None
    if dist is None:
        try:
            uid = unique_id()
        except Exception as e:  # noqa: BLE001
            say("mocap_comm_unique_id", e)
    else:
        box = torch.zeros(_abi.COMM_ID_BYTES, dtype=torch.uint8, device=flag_device)
        if rank == 0:
            try:
                box.copy_(torch.frombuffer(bytearray(unique_id()), dtype=torch.uint8))
            except Exception as e:  # noqa: BLE001 -- the broadcast below still runs (every rank is waiting in it) and carries zeros
                say("mocap_comm_unique_id", e)
        dist.broadcast(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        raw = bytes(box.cpu().numpy().tobytes())
        uid = raw if any(raw) else None
    if not agree(uid is not None):
        release()
        return False
    have = False
    try:
        comm_init(uid)
        have = True
    except Exception as e:  # noqa: BLE001
        say("mocap_comm_init", e)
    if not agree(have):
        if have:
            destroy()
        release()
        return False
    try:
        share()
        shared = True
    except Exception as e:  # noqa: BLE001
        say("mocap_comm_share", e)
        shared = False
    if not agree(shared):
        destroy()
        release()
        return False
    return True


# per-time-step status codes of the correspondence kernel (include/mocap_hip.h MOCAP_CORR_E_*)
CORR_STATUS = {
    -2: "more candidate groups than the correspondence kernel's capacity (MOCAP_CORR_E_GROUPS): more than 16 candidates for "
        "one (root, camera), more than max_groups groups for one root (raise max_groups), or more groups in the whole time "
        "step than max(2 * max_groups, 8192) (raise it with MocapContext.set_tuning('corr_step_groups', n))",
    -3: "a camera holds more image points than the tracker's max_points (MOCAP_CORR_E_TRUNCATED): raise max_points",
    -4: "a camera's blob stage exceeded an internal capacity (MOCAP_CORR_E_BLOB; its record count holds the MOCAP_BLOB_E_* code)",
}


class CapacityError(RuntimeError):
    """A time step has no result because a fixed capacity was exceeded.  The reference has no such limits
    (lib/Helpers.py:191,203-245), so the batched path never passes a shortened point list for a result: it fails."""

    def __init__(self, step, code):
        super().__init__(f"time step {step}: {CORR_STATUS.get(code, 'status %d' % code)}")
        self.step, self.code = step, code


def check_status(n_roots, first_step=0):
    """n_roots: host array of the per-time-step root counts of a batch.  Raises CapacityError for the first failed
    time step."""
    n_roots = np.asarray(n_roots)
    bad = np.flatnonzero(n_roots < 0)
    if len(bad):
        raise CapacityError(first_step + int(bad[0]), int(n_roots[bad[0]]))
    return n_roots


class _Lane:
    """One set of per-batch buffers (context scratch, centroid records, outputs) and the HIP stream it works on."""

    def __init__(self, ctx, records, stream):
        self.ctx, self.records, self.stream, self.out = ctx, records, stream, None
        self.gathered = None # all ranks' records (collective == "rccl")
        self.staging = None  # device copy of a batch that arrived in host memory
        self.gray = None     # gray frames of a batch that arrived as raw Bayer frames


class BatchTracker:
    """Frames -> 3-D marker positions for `steps_per_rank` time steps per call on this rank.

    `depth` > 1 software-pipelines consecutive batches: batch k runs on HIP stream k % depth with its own scratch
    buffers, so the latency-bound tail of one batch (border following, correspondence) overlaps the HBM-bound head of
    the next (the streaming scan).  Results of a batch are complete once its stream (or the device) is synchronised."""

    def __init__(self, K, dist, R, t, F, width, height, steps_per_rank, world=1, rank=0, device=0, group=None,
                 max_points=32, max_groups=4096, depth=1, bayer_pattern=None, gray_shift=GRAY_SHIFT, collective="auto",
                 force_collective=False):
        """collective: how the centroid records are exchanged when world > 1 --
             "rccl"  mocap_allgather_centroids: ncclAllGather called by the library on the batch's own HIP stream; ONE
                     communicator per rank, shared by the batches in flight (the library chains their all-gathers with
                     an event); its unique id travels through torch.distributed once, at set-up (negotiate_rccl);
             "torch" torch.distributed (all_gather_into_tensor on nccl = RCCL, or gloo through the host);
             "auto"  "rccl" when the process group's backend is nccl, else "torch".
           force_collective: run the exchange even with world == 1 (a one-rank all-gather; tests)."""
        self.n_cam = len(K)
        self.T = int(steps_per_rank)
        self.world, self.rank, self.group = world, rank, group
        self.t_total = self.T * world
        self.max_points, self.max_groups = max_points, max_groups
        # raw sensor frames: the camera loop's Bayer -> BGR -> gray (RealtimeTracking_FLIR.py:103-104) runs first, on the
        # batch's own stream and into the batch's own gray buffer (pattern 0..3 = BG, GB, RG, GR; None = gray frames)
        self.bayer_pattern, self.gray_shift = bayer_pattern, gray_shift
        self.segs = shard_plan(self.n_cam, self.T, world, rank)
        local_cams = sorted({c for c, _, _ in self.segs})
        self.slot_of = {c: i for i, c in enumerate(local_cams)}
        n_slots = self.n_cam if world == 1 else len(local_cams)
        self.per = self.n_cam * self.T
        # centroid record of one image: count, pad, (x, y) x max_points -- correspondence reads at most max_points points
        # per camera, so longer records would only inflate the all-gather
        self.rec_ints = 2 + 2 * max_points
        self.lanes = []
        for d in range(max(1, int(depth))):
            ctx = MocapContext(width, height, n_slots, device)
            if world == 1:
                for c in range(self.n_cam):
                    ctx.set_undistort(c, K[c], dist[c], warn_dense=d == 0)  # the lanes hold the same tables: one warning
            else:
                for c in local_cams:
                    ctx.set_undistort(self.slot_of[c], K[c], dist[c], warn_dense=d == 0)
            ctx.set_cameras(K, dist, R, t)
            ctx.set_fundamentals(F)
            records = torch.zeros((self.per, self.rec_ints), dtype=torch.int32, device=ctx.device)
            stream = torch.cuda.Stream(device=ctx.device) if depth > 1 else None
            self.lanes.append(_Lane(ctx, records, stream))
        self._k = 0
        self._cur = self.lanes[0]
        self.force_collective = bool(force_collective)
        self.collective = None
        if world > 1 or self.force_collective:
            self.collective = self._setup_collective(collective)

    def _setup_collective(self, collective):
        import torch.distributed as dist
        have_pg = dist.is_available() and dist.is_initialized()
        if collective == "auto":
            collective = "rccl" if (not have_pg and self.world == 1) or (have_pg and dist.get_backend(self.group) == "nccl") \
                else "torch"
        if collective == "torch":
            return "torch"
        assert collective == "rccl", collective
        from .engine import comm_available, comm_unique_id
        dev = self.lanes[0].ctx.device
        on_gpu = have_pg and dist.get_backend(self.group) == "nccl"
        flag_dev = dev if on_gpu else "cpu"
        first = self.lanes[0].ctx

        def prepare():  # local: RCCL loadable here, room for every lane's gathered records
            comm_available()
            for lane in self.lanes:
                lane.gathered = torch.empty((self.world * self.per, self.rec_ints), dtype=torch.int32, device=dev)

        def release():
            for lane in self.lanes:
                lane.gathered = None

        def share():    # local: the other batches in flight use the same communicator
            for lane in self.lanes[1:]:
                lane.ctx.comm_share(first)

        def destroy():
            for lane in self.lanes:
                lane.ctx.comm_destroy()

        ok = negotiate_rccl(self.world, self.rank, prepare=prepare, unique_id=comm_unique_id,
                            comm_init=lambda uid: first.comm_init(uid, self.rank, self.world), share=share, destroy=destroy,
                            release=release, dist=dist if (self.world > 1 and have_pg) else None, group=self.group, flag_device=flag_dev)
        return "rccl" if ok else "torch"

    # the buffers of the batch most recently submitted
    @property
    def ctx(self):
        return self._cur.ctx

    @property
    def records(self):
        return self._cur.records

    @property
    def out(self):
        return self._cur.out

    @out.setter
    def out(self, v):
        self._cur.out = v

    def local_image_list(self):
        """(camera, global time step) of every local image, in the order `step` expects the frames."""
        if self.world == 1:
            return [(c, t) for t in range(self.T) for c in range(self.n_cam)]  # time-major [T][C]
        return [(c, t) for c, t0, t1 in self.segs for t in range(t0, t1)]

    def extract(self, frames):
        """Stage A on this rank's block.  frames: uint8 [per, H, W] on the GPU, ordered as local_image_list(); raw Bayer
        frames when the tracker was built with `bayer_pattern` (their gray version goes to the batch's own buffer).
        Fills and returns self.records ([per, REC] int32, one centroid record per image)."""
        ctx = self.ctx
        kw = {}
        if self.bayer_pattern is not None:
            lane = self._cur
            if lane.gray is None or lane.gray.shape != frames.shape:
                lane.gray = torch.empty(frames.shape, dtype=torch.uint8, device=ctx.device)
            kw = {"bayer_pattern": self.bayer_pattern, "gray_shift": self.gray_shift}
        if self.world == 1:
            if kw:
                kw["gray"] = self._cur.gray
            ctx.blob_centroids(frames, cam_mod=self.n_cam, max_blobs=self.max_points, records=self.records, **kw)
            return self.records
        o = 0
        for c, t0, t1 in self.segs:
            n = t1 - t0
            if kw:
                kw["gray"] = self._cur.gray[o:o + n]
            ctx.blob_centroids(frames[o:o + n], cam_mod=1, slot_base=self.slot_of[c], max_blobs=self.max_points,
                               records=self.records[o:o + n], **kw)
            o += n
        return self.records

    def triangulate(self, gathered):
        """Stage B for this rank's time steps [rank * T, (rank + 1) * T) from the records of all cameras
        (world == 1: self.records, time-major; else the all-gathered records, camera-major)."""
        ctx = self.ctx
        if self.world == 1:
            self.out = ctx.correspond_records(gathered, self.T, self.n_cam, t0=0, stride_t=self.n_cam, stride_c=1,
                                              P=self.max_points, max_groups=self.max_groups, out=self.out)
        else:
            self.out = ctx.correspond_records(gathered, self.T, self.n_cam, t0=self.rank * self.T, stride_t=1,
                                              stride_c=self.t_total, P=self.max_points, max_groups=self.max_groups,
                                              out=self.out)
        return self.out

    def step(self, frames):
        """One pass of the hot path over this rank's block: extract -> (all-gather) -> triangulate.  Returns the
        correspondence outputs (dict of device tensors, see MocapContext.correspond) for this rank's time steps.
        With depth > 1 the work is queued on the batch's own stream and the call returns at once."""
        lane = self.lanes[self._k % len(self.lanes)]
        self._k += 1
        self._cur = lane
        if lane.stream is None:
            return self._run(self._resident(lane, frames))
        lane.stream.wait_stream(torch.cuda.current_stream())  # the frames were produced on the caller's stream
        if frames.is_cuda:
            frames.record_stream(lane.stream)  # the caller may drop them right after this call: keep the memory until the lane is done
        with torch.cuda.stream(lane.stream):
            return self._run(self._resident(lane, frames))

    def _resident(self, lane, frames):
        """Frames in (pinned) host memory are uploaded into the batch's own staging buffer on its stream, so the copy
        of one batch overlaps the kernels of the others; frames already on the GPU are used in place."""
        if frames.is_cuda:
            return frames
        if lane.staging is None or lane.staging.shape != frames.shape:
            lane.staging = torch.empty(frames.shape, dtype=torch.uint8, device=lane.ctx.device)
        lane.staging.copy_(frames, non_blocking=True)
        return lane.staging

    def _run(self, frames):
        records = self.extract(frames)
        if self.collective == "rccl":    # [C * T_total, REC] camera-major when world > 1
            gathered = self._cur.ctx.allgather_centroids(records, self._cur.gathered)
        else:
            gathered = allgather_records(records, self.world, self.group, force=self.force_collective)
        return self.triangulate(gathered)

    def synchronize(self):
        """Wait for every batch submitted so far."""
        for lane in self.lanes:
            if lane.stream is not None:
                lane.stream.synchronize()
            else:
                torch.cuda.current_stream(lane.ctx.device).synchronize()

    def finish(self, out=None, first_step=0):
        """Wait for the batch that produced `out` (default: the most recent one) and return its per-time-step root
        counts as a host array.  Raises CapacityError when a time step failed (more image points in a camera than
        max_points, a blob-stage capacity error, too many candidate groups): a batch is never answered with shortened
        point lists."""
        lane = self._cur
        if out is not None:
            lane = next((l for l in self.lanes if l.out is out), lane)
        out = lane.out
        if lane.stream is not None:
            lane.stream.synchronize()
        else:
            torch.cuda.current_stream(lane.ctx.device).synchronize()
        return check_status(out["n"].cpu().numpy(), first_step)

    def profile(self, on=True):
        """HIP-event timing of the kernels of every batch in flight (MocapContext.profile)."""
        for lane in self.lanes:
            lane.ctx.profile(on)

    def profile_read(self):
        """Sums of MocapContext.profile_read over the batches in flight."""
        total = {}
        for lane in self.lanes:
            for k, v in lane.ctx.profile_read().items():
                total[k] = total.get(k, 0) + v
        return total


def scene_arrays(scene):
    """(K, dist, R, t, F) arrays of a synth.Scene"""
    C = scene.n_cam
    return (np.stack([scene.K] * C), np.stack([scene.dist] * C), np.stack([p["R"] for p in scene.poses]),
            np.stack([p["t"] for p in scene.poses]), np.stack(scene.Fs))
