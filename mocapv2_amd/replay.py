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

    def run(self, frames, send=None):
        """Generator over time steps.  frames: uint8 [T, C, H, W] NumPy array or torch tensor (host or device).
        Yields dicts: object_points [<= obj_count+1, 3] (or shape (0,)), image_points [roots, C, 2] (or (0,)),
        message (bytes).  `send(bytes)` is called per time step when given."""
        T = frames.shape[0]
        assert frames.shape[1:] == (self.n_cam, self.height, self.width), frames.shape
        dev = self.tracker.ctx.device
        pending = []  # batches submitted and not yet read back: (first time step, time steps, outputs); at most depth - 1 wait here
        for b0 in range(0, T, self.batch):
            chunk = frames[b0:b0 + self.batch]
            nb = chunk.shape[0]
            if isinstance(chunk, np.ndarray):
                chunk = torch.from_numpy(np.ascontiguousarray(chunk))
            chunk = chunk.to(dev, non_blocking=True)
            if nb < self.batch:  # pad the last batch with black frames (they produce no points)
                pad = torch.zeros((self.batch - nb,) + tuple(chunk.shape[1:]), dtype=torch.uint8, device=dev)
                chunk = torch.cat([chunk, pad], dim=0)
            out = self.tracker.step(chunk.reshape(self.batch * self.n_cam, self.height, self.width).contiguous())
            pending.append((b0, nb, out))
            if len(pending) >= self.depth:  # the oldest batch's lane is the next one to be reused: read it back first
                yield from self._emit(*pending.pop(0), send)
        while pending:
            yield from self._emit(*pending.pop(0), send)

    def _emit(self, b0, nb, out, send):
        n = self.tracker.finish(out, first_step=b0)  # raises CapacityError: no time step is answered from shortened lists
        xyz = out["xyz"].cpu().numpy()
        grp = out["grp"].cpu().numpy()
        order = out["order"].cpu().numpy()
        for s in range(nb):
            k = int(n[s])
            if k == 0:
                obj, img = np.array([]), np.array([])
            else:
                obj = self._select(xyz[s], order[s], k)
                img = grp[s, :k].astype(np.int64)
            if len(obj) > 0:
                self.point = [0, 0, 0, 0] + list(obj[0])  # :184-185
            msg = tracker_message(self.point)
            if send is not None:
                send(msg)
            yield {"object_points": obj, "image_points": img, "message": msg}


__all__ = ["ReplayTracker", "tracker_message", "unpack_tracker_message", "OBJ_COUNT", "MocapContext"]
