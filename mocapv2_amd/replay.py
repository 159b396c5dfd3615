"""Headless replay tracker: the reference's `track_points` + `track` loops (RealtimeTracking_FLIR.py:95-143, 157-209)
fed from recorded / synthetic frames instead of PySpin cameras, in batches through the HIP path.

What the reference does per time step, and what is kept here:
  * every camera thread runs `_find_dot` on its frame and queues the image points (:105-112);
  * `track` pairs the newest entries of the queues (:177-180), calls
    `find_point_correspondance_and_object_points(image_points, camera_poses, 4)` (:181) and sends
    `msgpack.packb({"tracker1": [0, 0, 0, 0, x, y, z]}, use_bin_type=True)` over TCP 127.0.0.1:5000 (:183-188); when
    no object point came out, the previous message is sent again (`point` keeps its value; it starts as eight zeros,
    :171).
Here frames of the same time step are paired by index (a recording has no queue races), the socket is the caller's
business (`send=` callback; networking is outside the path), and the message bytes are the reference's.
"""
import numpy as np
import torch

from .engine import GRAY_SHIFT, MocapContext
from .pipeline import BatchTracker

OBJ_COUNT = 4  # RealtimeTracking_FLIR.py:181


def tracker_message(point):
    """Wire format of one tracker update (RealtimeTracking_FLIR.py:186-187): msgpack map {"tracker1": point}."""
    import msgpack
    return msgpack.packb({"tracker1": [float(v) if isinstance(v, (float, np.floating)) else v for v in point]},
                         use_bin_type=True)


def unpack_tracker_message(data):
    import msgpack
    return msgpack.unpackb(data, raw=False)["tracker1"]


class ReplayTracker:
    """frames [T, C, H, W] (gray, or raw Bayer with `bayer_pattern`) -> per time step (object_points, image_points, message) exactly as `track` would produce
    them from the same detections.  `batch` time steps go through the GPU at once."""

    def __init__(self, K, dist, R, t, F, width, height, batch=64, obj_count=OBJ_COUNT, device=0, max_points=32,
                 max_groups=4096, bayer_pattern=None, gray_shift=GRAY_SHIFT, depth=1):
        """depth > 1: that many batches are in flight on HIP streams of their own (BatchTracker's software pipelining): while
        the results of one batch are read back and turned into messages, the next ones are already on the GPU.  The time steps
        come out in order either way."""
        self.n_cam = len(K)
        self.batch = int(batch)
        self.obj_count = obj_count
        self.width, self.height = width, height
        self.depth = max(1, int(depth))
        self.tracker = BatchTracker(K, dist, R, t, F, width, height, self.batch, device=device, max_points=max_points,
                                    max_groups=max_groups, bayer_pattern=bayer_pattern, gray_shift=gray_shift, depth=self.depth)
        self.point = [0, 0, 0, 0, 0, 0, 0, 0]  # RealtimeTracking_FLIR.py:171 (eight zeros until the first detection)
        # raw sensor frames: the camera loop's cvtColor(BAYER_GR2BGR) + cvtColor(BGR2GRAY) (:103-104) run on the GPU first;
        # bayer_pattern 0..3 = BG, GB, RG, GR (the reference: 3), None = the frames are gray already
        self.bayer_pattern, self.gray_shift = bayer_pattern, gray_shift

    def _select(self, xyz, order, n):
        idx = order[:n]
        if not self.obj_count > len(idx):  # lib/Helpers.py:275-278
            idx = idx[: self.obj_count + 1]
        return xyz[idx]

    def _submit(self, frames):
        """Generator over the batches of `frames` (one array / tensor [T, C, H, W], or a list of such pieces), each submitted to the
        GPU: (first time step, time steps, outputs), at most `depth` of them in flight, oldest first."""
        dev = self.tracker.ctx.device
        pending = []  # batches submitted and not yet read back; at most depth - 1 wait here
        parts = frames if isinstance(frames, (list, tuple)) else [frames]  # a recording in several pieces: one after the other
        t_base = 0
        for part in parts:
            assert part.shape[1:] == (self.n_cam, self.height, self.width), part.shape
            for p0 in range(0, part.shape[0], self.batch):
                chunk = part[p0:p0 + self.batch]
                nb = chunk.shape[0]
                if isinstance(chunk, np.ndarray):
                    chunk = torch.from_numpy(np.ascontiguousarray(chunk))
                # depth 1: the plain blocking copy.  Deeper pipelines upload asynchronously (a pinned caller tensor is then read
                # later: the caller must leave that piece of the frames untouched until the batch's results have been yielded)
                chunk = chunk.to(dev, non_blocking=self.depth > 1)
                if nb < self.batch:  # pad the last batch with black frames (they produce no points)
                    pad = torch.zeros((self.batch - nb,) + tuple(chunk.shape[1:]), dtype=torch.uint8, device=dev)
                    chunk = torch.cat([chunk, pad], dim=0)
                out = self.tracker.step(chunk.reshape(self.batch * self.n_cam, self.height, self.width).contiguous())
                pending.append((t_base + p0, nb, out))
                if len(pending) >= self.depth:  # the oldest batch's lane is the next one to be reused: read it back first
                    yield pending.pop(0)
            t_base += part.shape[0]
        while pending:
            yield pending.pop(0)

    def _collect(self, b0, nb, out):
        """One batch read back (one device-to-host copy per output) and turned into what `track` produces, for all of its time
        steps at once: the `obj_count + 1` selection of lib/Helpers.py:274-279, the image points, and the message bytes -- built
        in bulk (batch_messages), the previous message repeated for time steps without a point (RealtimeTracking_FLIR.py:181-188)."""
        n = self.tracker.finish(out, first_step=b0)[:nb]  # raises CapacityError: no time step is answered from shortened lists
        xyz = out["xyz"].cpu().numpy()[:nb]
        grp = out["grp"].cpu().numpy()[:nb].astype(np.int64)
        order = out["order"].cpu().numpy()[:nb]
        oc = self.obj_count
        kept = np.where(n >= oc, np.minimum(n, oc + 1), n)  # `if not obj_count > len(...)`: lib/Helpers.py:275-278
        width = min(oc + 1, xyz.shape[1])
        sel = np.clip(order[:, :width], 0, xyz.shape[1] - 1)
        obj = xyz[np.arange(nb)[:, None], sel]  # [nb, <= obj_count + 1, 3]; rows beyond kept[s] are not results
        has = kept > 0
        msgs = batch_messages(obj[:, 0] if width else np.zeros((nb, 3)), has, tracker_message(self.point))
        if has.any():
            self.point = [0, 0, 0, 0] + list(obj[np.flatnonzero(has)[-1], 0])  # :184-185
        return {"first_step": b0, "n_steps": nb, "n_roots": n, "kept": kept, "object_points": obj, "image_points": grp, "messages": msgs}

    def run_batches(self, frames, send_many=None):
        """Generator over BATCHES of time steps: dicts with first_step, n_steps, n_roots [n], kept [n] (object points per time
        step), object_points [n, <= obj_count + 1, 3] (rows beyond kept[s] unused), image_points [n, P, C, 2] int64 (rows
        beyond n_roots[s] unused), messages (list of n bytes objects).  `send_many(list_of_bytes)` is called once per batch when
        given.  The bulk form of run(): no Python work per time step."""
        for b0, nb, out in self._submit(frames):
            res = self._collect(b0, nb, out)
            if send_many is not None:
                send_many(res["messages"])
            yield res

    def run(self, frames, send=None):
        """Generator over time steps.  frames: uint8 [T, C, H, W] NumPy array or torch tensor (host or device).
        Yields dicts: object_points [<= obj_count+1, 3] (or shape (0,)), image_points [roots, C, 2] (or (0,)),
        message (bytes).  `send(bytes)` is called per time step when given."""
        empty = np.array([])
        for res in self.run_batches(frames):
            n, kept, obj, img, msgs = res["n_roots"], res["kept"], res["object_points"], res["image_points"], res["messages"]
            for s in range(res["n_steps"]):
                if send is not None:
                    send(msgs[s])
                k = int(n[s])
                yield {"object_points": obj[s, :kept[s]] if k else empty, "image_points": img[s, :k] if k else empty, "message": msgs[s]}


_MSG_HEAD = bytes([0x81, 0xa8]) + b"tracker1" + bytes([0x97, 0, 0, 0, 0])  # map of 1, fixstr(8), array of 7, four zero ints


def batch_messages(first_points, has, carry):
    """The tracker messages of a run of time steps, built without a Python loop over them.  first_points [n, 3] float64 (the
    first object point of each time step), has [n] bool (the time step produced a point), carry: the message in force before
    the run.  Returns a list of n bytes objects: msgpack.packb({"tracker1": [0, 0, 0, 0, x, y, z]}, use_bin_type=True) for a
    time step with a point (a map of one entry, fixstr key, array of seven: four zero fixints and three float64, big endian),
    else the message of the last time step that had one (`point` keeps its value, RealtimeTracking_FLIR.py:181-188)."""
    n = len(has)
    L = len(_MSG_HEAD) + 27
    buf = np.empty((n, L), np.uint8)
    buf[:, :len(_MSG_HEAD)] = np.frombuffer(_MSG_HEAD, np.uint8)
    body = buf[:, len(_MSG_HEAD):].reshape(n, 3, 9)
    body[:, :, 0] = 0xcb
    body[:, :, 1:] = np.ascontiguousarray(first_points, np.float64).astype(">f8").view(np.uint8).reshape(n, 3, 8)
    src = np.maximum.accumulate(np.where(has, np.arange(n), -1))  # the time step whose message is in force
    raw = buf[np.maximum(src, 0)].tobytes()
    return [raw[i * L:(i + 1) * L] if src[i] >= 0 else carry for i in range(n)]


__all__ = ["ReplayTracker", "batch_messages", "tracker_message", "unpack_tracker_message", "OBJ_COUNT", "MocapContext"]
