"""Host side of the hot path: a thin object over the C-ABI context.

PyTorch is used only as the device allocator / stream provider (tensors are passed to the library as raw
pointers); no torch operator touches the data.  Mirrors what the reference's callers do around
lib.ImageOperations._find_dot and lib.Helpers.find_point_correspondance_and_object_points
(reference RealtimeTracking_FLIR.py:95-143,157-209) for whole batches of frames.
"""
import ctypes as C
import threading

import numpy as np
import torch

from . import _abi

MAX_BLOBS = 128       # centroid record capacity per camera image
# Fixed point of cv2.cvtColor(., COLOR_BGR2GRAY) on 8-bit images in the camera loop (RealtimeTracking_FLIR.py:104).
# The reference pins no OpenCV version (README.md:65), so `pip install opencv-python` gives a current 4.x, whose
# RGB2Gray<uchar> works with gray_shift = 15 (RY15 = 9798, GY15 = 19235, BY15 = 3735, + 2^14, >> 15;
# modules/imgproc/src/color.simd_helpers.hpp); OpenCV 2.x / 3.x used the 14-bit set (4899, 9617, 1868).  Both are
# implemented and tested; the default follows the version the reference's users get today.  oracle/check_against_cv2.py
# settles it wherever cv2 can be imported (it cannot in the build container).
GRAY_SHIFT = 15
REC_INTS = 2 + 2 * MAX_BLOBS  # int32 record: count, pad, xy[MAX_BLOBS][2]


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dbl(a, n):
    a = np.ascontiguousarray(a, np.float64).reshape(-1)
    assert a.size == n, (a.size, n)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


class MocapContext:
    """One GPU, one image geometry.  Not a singleton: one per camera thread or per process is fine."""

    def __init__(self, width, height, n_slots=1, device=0):
        self.lib = _abi.load()
        if not torch.cuda.is_available():
            raise RuntimeError("mocapv2_amd needs a ROCm GPU (torch.cuda.is_available() is False); no CPU fallback")
        self.device = torch.device("cuda", device)
        self.width, self.height, self.n_slots = int(width), int(height), int(n_slots)
        h = C.c_void_p()
        _abi.check(self.lib.mocap_ctx_create(device, self.width, self.height, self.n_slots, C.byref(h)))
        self._h = h
        self._cam_key = None
        self._f_key = None
        self._und_key = {}
        self._warned_dense = set()  # causes set_undistort has warned about
        self.identity = {}

    def close(self):
        if getattr(self, "_h", None):
            self.lib.mocap_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- set-up -------------------------------------------------------------------------------------------
    def set_blob_params(self, thresh=255 * 0.85, min_area=500.0, min_circ=0.5, ksize=5, median=5):
        p = _abi.BlobParams(ksize, median, thresh, min_area, min_circ)
        _abi.check(self.lib.mocap_set_blob_params(self._h, C.byref(p)))

    def set_tuning(self, name, value):
        """One performance switch of this context (mocap_set_tuning; names in include/mocap_hip.h).  None changes a result."""
        _abi.check(self.lib.mocap_set_tuning(self._h, name.encode(), int(value)))

    def set_undistort(self, slot, K, dist, warn_dense=True):
        K, Kp = _dbl(K, 9)
        d, dp = _dbl(dist, 5)
        key = K.tobytes() + d.tobytes()
        if self._und_key.get(slot) == key:
            return self.identity[slot]
        ident = C.c_int(0)
        _abi.check(self.lib.mocap_set_undistort(self._h, slot, Kp, dp, C.byref(ident)))
        self._und_key[slot] = key
        self.identity[slot] = bool(ident.value)
        if warn_dense:
            # Only a table that cannot take the sparse road is worth a warning: sparse_path is also False when a tuning switch
            # (general_filter, skip_dark=0) sends everything down the dense kernel on purpose.  Once per cause and context.
            info = self.undistort_info(slot)
            causes = tuple(k for k in ("compact_table", "early_out_provable") if not info["identity"] and not info[k])
            if causes and causes not in self._warned_dense:
                self._warned_dense.add(causes)
                import warnings
                warnings.warn(f"mocapv2_amd: undistort slot {slot}: the lens table is outside the bounds of the sparse path "
                              f"({', '.join(c + ' = False' for c in causes)}; {info}); its images are filtered by the dense kernel: "
                              "same results, several times slower on dark scenes", RuntimeWarning, stacklevel=2)
        return self.identity[slot]

    def undistort_info(self, slot=0):
        """What set_undistort found out about the slot's table (mocap_undistort_info): dict with identity, compact_table,
        early_out_provable, max_source_weight, sparse_path.  sparse_path False = the slot's images take the dense row
        pipeline (same results, several times slower on a dark scene)."""
        info = _abi.UndistortInfo()
        _abi.check(self.lib.mocap_undistort_info(self._h, int(slot), C.byref(info)))
        return {k: (bool(getattr(info, k)) if k != "max_source_weight" else int(info.max_source_weight)) for k, _ in info._fields_}

    def set_cameras(self, K, dist, R, t):
        n = len(K)
        K, Kp = _dbl(K, 9 * n)
        d, dp = _dbl(dist, 5 * n)
        R, Rp = _dbl(R, 9 * n)
        t, tp = _dbl(t, 3 * n)
        key = K.tobytes() + d.tobytes() + R.tobytes() + t.tobytes()
        if key != self._cam_key:
            _abi.check(self.lib.mocap_set_cameras(self._h, n, Kp, dp, Rp, tp))
            self._cam_key = key
        self.n_cam = n

    def set_fundamentals(self, F):
        F = np.ascontiguousarray(F, np.float64).reshape(-1, 9)
        key = F.tobytes()
        if key != self._f_key:
            _abi.check(self.lib.mocap_set_fundamentals(self._h, len(F), F.ctypes.data_as(C.POINTER(C.c_double))))
            self._f_key = key

    def sync(self):
        _abi.check(self.lib.mocap_sync(self._h, _stream()))

    # ---- blob stage ------------------------------------------------------------------------------------------
    def _frames(self, frames):
        assert frames.is_cuda and frames.dtype == torch.uint8
        assert frames.shape[-2:] == (self.height, self.width), (frames.shape, self.height, self.width)
        assert frames.stride(-1) == 1
        n = int(np.prod(frames.shape[:-2])) if frames.dim() > 2 else 1
        pitch = frames.stride(-2)
        flat = frames.reshape(n, self.height, self.width) if frames.is_contiguous() else frames
        assert flat.dim() == 3
        stride = flat.stride(0) if n > 1 else pitch * self.height
        return flat, n, stride, pitch

    def blob_centroids(self, frames, cam_mod=1, slot_base=0, max_blobs=MAX_BLOBS, records=None, bayer_pattern=None,
                       gray_shift=GRAY_SHIFT, gray=None):
        """_find_dot over uint8 frames [..., H, W] resident on the GPU (image n uses undistort slot
        slot_base + n % cam_mod).  Results land in centroid records, int32 [n, 2 + 2*max_blobs]:
        record[0] = number of image points, record[2:] = (cx, cy) pairs in the reference's contour order.
        Returns the records tensor (use record_views for xy / count views).
        bayer_pattern 0..3 (BG, GB, RG, GR): the frames are raw sensor frames; the camera loop's cvtColor pair
        (RealtimeTracking_FLIR.py:103-104) runs first, into `gray` (same shape as frames; allocated when None)."""
        flat, n, stride, pitch = self._frames(frames)
        rec_ints = 2 + 2 * max_blobs
        if records is None:
            records = torch.empty((n, rec_ints), dtype=torch.int32, device=self.device)
        assert records.is_contiguous() and records.shape == (n, rec_ints) and records.dtype == torch.int32
        xy_ptr = C.c_void_p(records.data_ptr() + 8)
        if bayer_pattern is None:
            _abi.check(self.lib.mocap_blob_centroids(self._h, _ptr(flat), n, cam_mod, slot_base, stride, pitch, xy_ptr, rec_ints,
                                                     _ptr(records), rec_ints, max_blobs, _stream()))
            return records
        if gray is None:
            gray = torch.empty_strided(flat.shape, flat.stride(), dtype=torch.uint8, device=self.device)
        assert gray.dtype == torch.uint8 and gray.is_cuda and gray.reshape(flat.shape).stride() == flat.stride()
        _abi.check(self.lib.mocap_blob_centroids_bayer(self._h, _ptr(flat), _ptr(gray), n, cam_mod, slot_base, stride, pitch,
                                                       bayer_pattern, gray_shift, xy_ptr, rec_ints, _ptr(records), rec_ints,
                                                       max_blobs, _stream()))
        return records

    @staticmethod
    def record_views(records):
        """(xy [n, max_blobs, 2], count [n]) views of a records tensor"""
        n, rec_ints = records.shape
        return records[:, 2:].reshape(n, (rec_ints - 2) // 2, 2), records[:, 0]

    def filter_mask(self, frames, cam_mod=1, slot_base=0, mask=None):
        flat, n, stride, pitch = self._frames(frames)
        wpr = (self.width + 31) // 32
        if mask is None:
            mask = torch.zeros((n, self.height, wpr), dtype=torch.int32, device=self.device)
        _abi.check(self.lib.mocap_filter_mask(self._h, _ptr(flat), n, cam_mod, slot_base, stride, pitch, _ptr(mask), _stream()))
        return mask

    def contours_from_mask(self, mask, max_blobs=MAX_BLOBS, debug_cap=0):
        n = mask.shape[0]
        xy = torch.empty((n, max_blobs, 2), dtype=torch.int32, device=self.device)
        cnt = torch.empty((n,), dtype=torch.int32, device=self.device)
        dbg = dbg_n = None
        if debug_cap:
            dbg = torch.zeros((n, debug_cap, C.sizeof(_abi.Contour)), dtype=torch.uint8, device=self.device)
            dbg_n = torch.zeros((n,), dtype=torch.int32, device=self.device)
        _abi.check(self.lib.mocap_contours_from_mask(self._h, _ptr(mask), n, _ptr(xy), 2 * max_blobs, _ptr(cnt), 1, max_blobs,
                                                     _ptr(dbg), _ptr(dbg_n), debug_cap, _stream()))
        if not debug_cap:
            return xy, cnt
        raw = dbg.cpu().numpy()
        counts = dbg_n.cpu().numpy()
        recs = []
        for i in range(n):
            arr = (_abi.Contour * debug_cap).from_buffer_copy(raw[i].tobytes())
            recs.append([{k: getattr(arr[j], k) for k, _ in _abi.Contour._fields_} for j in range(min(counts[i], debug_cap))])
        return xy, cnt, recs

    def image_filter(self, img, order=0, slot=-1):
        """image_filter_gpu (order 0) / image_filter_cpu (order 1) on one device image -> {0,255} device image."""
        assert img.is_cuda and img.dtype == torch.uint8 and img.shape == (self.height, self.width)
        out = torch.empty_like(img, memory_format=torch.contiguous_format)
        _abi.check(self.lib.mocap_image_filter_u8(self._h, _ptr(img), _ptr(out), img.stride(0), out.stride(0), order, slot,
                                                  _stream()))
        return out

    def undistort(self, img, slot=0):
        assert img.is_cuda and img.dtype == torch.uint8 and img.shape == (self.height, self.width)
        out = torch.empty_like(img, memory_format=torch.contiguous_format)
        _abi.check(self.lib.mocap_undistort_u8(self._h, slot, _ptr(img), _ptr(out), img.stride(0), out.stride(0), _stream()))
        return out

    def box_blur(self, img, ksize=5):
        assert img.is_cuda and img.dtype == torch.uint8 and img.dim() == 2
        out = torch.empty_like(img, memory_format=torch.contiguous_format)
        _abi.check(self.lib.mocap_box_blur_u8(self._h, _ptr(img), _ptr(out), img.shape[0], img.shape[1], img.stride(0),
                                              out.stride(0), ksize, _stream()))
        return out

    def demosaic(self, bayer):
        assert bayer.is_cuda and bayer.dtype == torch.uint8 and bayer.dim() == 2
        out = torch.empty(bayer.shape + (3,), dtype=torch.uint8, device=self.device)
        _abi.check(self.lib.mocap_demosaic_u8(self._h, _ptr(bayer), _ptr(out), bayer.shape[0], bayer.shape[1],
                                              bayer.stride(0), _stream()))
        return out

    def bayer_gray(self, bayer, pattern=3, gray_shift=GRAY_SHIFT, out=None):
        """Raw Bayer frames uint8 [H, W] or [n, H, W] on the GPU -> gray frames of the same shape:
        cv2.cvtColor(cv2.cvtColor(raw, COLOR_BAYER_GR2BGR), COLOR_BGR2GRAY) of the reference's camera loop
        (RealtimeTracking_FLIR.py:103-104) in one pass.  pattern 0..3 = BG, GB, RG, GR."""
        assert bayer.is_cuda and bayer.dtype == torch.uint8 and bayer.dim() in (2, 3) and bayer.stride(-1) == 1
        b3 = bayer if bayer.dim() == 3 else bayer.unsqueeze(0)
        n, H, W = b3.shape
        out = torch.empty((n, H, W), dtype=torch.uint8, device=bayer.device) if out is None else out.reshape(n, H, W)
        assert out.is_contiguous() and out.dtype == torch.uint8
        _abi.check(self.lib.mocap_bayer_gray_u8(self._h, _ptr(b3), _ptr(out), n, H, W, b3.stride(1), W,
                                                b3.stride(0) if n > 1 else H * b3.stride(1), H * W, pattern, gray_shift,
                                                _stream()))
        return out if bayer.dim() == 3 else out[0]

    # ---- geometry stage ----------------------------------------------------------------------------------------
    def _corr_out(self, T, P, Cn):
        dev = self.device
        return {"xyz": torch.empty((T, P, 3), dtype=torch.float64, device=dev),
                "err": torch.empty((T, P), dtype=torch.float64, device=dev),
                "grp": torch.empty((T, P, Cn, 2), dtype=torch.float64, device=dev),
                "root": torch.empty((T, P), dtype=torch.int32, device=dev),
                "order": torch.empty((T, P), dtype=torch.int32, device=dev),
                "n": torch.empty((T,), dtype=torch.int32, device=dev)}

    def correspond(self, pts, counts, cutoff=10.0, max_groups=4096, out=None):
        """pts [T, C, P, 2] (int32 or float64), counts [T, C] int32, both on the GPU -> dict of device tensors."""
        assert pts.is_cuda and pts.is_contiguous() and counts.is_contiguous() and counts.dtype == torch.int32
        T, Cn, P, _ = pts.shape
        f64 = pts.dtype == torch.float64
        assert f64 or pts.dtype == torch.int32
        out = out or self._corr_out(T, P, Cn)
        _abi.check(self.lib.mocap_correspond(self._h, _ptr(pts), Cn * P * 2, P * 2, _ptr(counts), Cn, 1, int(f64), T, Cn, P,
                                             cutoff, max_groups, _ptr(out["xyz"]), _ptr(out["err"]), _ptr(out["grp"]),
                                             _ptr(out["root"]), _ptr(out["order"]), _ptr(out["n"]), _stream()))
        return out

    def correspond_records(self, records, T, Cn, t0=0, stride_t=None, stride_c=None, P=None, cutoff=10.0, max_groups=4096,
                           out=None):
        """Correspondence + triangulation straight from centroid records (int32 [..., 2 + 2*max_blobs]) laid out so
        that the record of (time step t0 + t, camera c) sits stride_t*t + stride_c*c records after records[t0 * ...].
        Default layout: records [T_total, C, rec] (time-major).  P limits the points read per camera."""
        rec_ints = records.shape[-1]
        flat = records.reshape(-1, rec_ints)
        stride_t = Cn if stride_t is None else stride_t
        stride_c = 1 if stride_c is None else stride_c
        P = P or min(255, (rec_ints - 2) // 2)
        out = out or self._corr_out(T, P, Cn)
        base = flat.data_ptr() + 4 * rec_ints * stride_t * t0
        _abi.check(self.lib.mocap_correspond(self._h, C.c_void_p(base + 8), rec_ints * stride_t, rec_ints * stride_c,
                                             C.c_void_p(base), rec_ints * stride_t, rec_ints * stride_c, 0, T, Cn, P,
                                             cutoff, max_groups, _ptr(out["xyz"]), _ptr(out["err"]), _ptr(out["grp"]),
                                             _ptr(out["root"]), _ptr(out["order"]), _ptr(out["n"]), _stream()))
        return out

    def epipolar_scores(self, roots, cand, f_index=0, with_lines=False):
        """The scoring step of the correspondence search for one camera pair (reference lib/Helpers.py:205-220): roots [R, 2]
        camera-0 points, cand [N, 2] points of camera f_index + 1 (host arrays, integer or float) -> distances [R, N] float64
        (and the float32 lines [R, 3]) as host arrays.  Requires set_fundamentals."""
        roots, cand = np.asarray(roots), np.asarray(cand)
        f64 = not (roots.dtype.kind in "iu" and cand.dtype.kind in "iu")
        dt = np.float64 if f64 else np.int32
        d_r = torch.from_numpy(np.ascontiguousarray(roots, dt).reshape(-1, 2)).to(self.device)
        d_c = torch.from_numpy(np.ascontiguousarray(cand, dt).reshape(-1, 2)).to(self.device)
        dist = torch.empty((d_r.shape[0], d_c.shape[0]), dtype=torch.float64, device=self.device)
        lines = torch.empty((d_r.shape[0], 3), dtype=torch.float32, device=self.device) if with_lines else None
        _abi.check(self.lib.mocap_epipolar_scores(self._h, _ptr(d_r), d_r.shape[0], _ptr(d_c), d_c.shape[0], int(f64), int(f_index),
                                                  _ptr(dist), _ptr(lines), _stream()))
        return (dist.cpu().numpy(), lines.cpu().numpy()) if with_lines else dist.cpu().numpy()

    def triangulate_batch(self, pts, valid, compact_k=True):
        """pts [N, C, 2] float64 host array, valid [N, C] -> (xyz [N,3], ok [N]) host arrays."""
        pts = np.ascontiguousarray(pts, np.float64)
        N, Cn = pts.shape[:2]
        d_pts = torch.from_numpy(pts).to(self.device)
        d_val = torch.from_numpy(np.ascontiguousarray(valid, np.uint8).reshape(N, Cn)).to(self.device)
        xyz = torch.zeros((N, 3), dtype=torch.float64, device=self.device)
        ok = torch.zeros((N,), dtype=torch.int32, device=self.device)
        _abi.check(self.lib.mocap_triangulate_batch(self._h, _ptr(d_pts), _ptr(d_val), N, Cn, int(compact_k), _ptr(xyz),
                                                    _ptr(ok), _stream()))
        return xyz.cpu().numpy(), ok.cpu().numpy()

    def reproject_batch(self, pts, valid, xyz, compact_k=True):
        pts = np.ascontiguousarray(pts, np.float64)
        N, Cn = pts.shape[:2]
        d_pts = torch.from_numpy(pts).to(self.device)
        d_val = torch.from_numpy(np.ascontiguousarray(valid, np.uint8).reshape(N, Cn)).to(self.device)
        d_xyz = torch.from_numpy(np.ascontiguousarray(xyz, np.float64).reshape(N, 3)).to(self.device)
        mse = torch.zeros((N,), dtype=torch.float64, device=self.device)
        ok = torch.zeros((N,), dtype=torch.int32, device=self.device)
        _abi.check(self.lib.mocap_reproject_batch(self._h, _ptr(d_pts), _ptr(d_val), _ptr(d_xyz), N, Cn, int(compact_k),
                                                  _ptr(mse), _ptr(ok), _stream()))
        return mse.cpu().numpy(), ok.cpu().numpy()

    def ba_problem(self, pts, valid=None):
        """Bundle-adjustment residuals with the image points resident on the GPU (see BAProblem)."""
        return BAProblem(self, pts, valid)

    # ---- the exchange step (one all-gather of centroid records, RCCL over xGMI) -----------------------------------
    def comm_init(self, unique_id, rank, world):
        """Collective over all ranks: create this context's RCCL communicator from the bytes rank 0 got from
        `comm_unique_id()` (handed around by the host, e.g. through torch.distributed)."""
        buf = (C.c_char * _abi.COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        _abi.check(self.lib.mocap_comm_init(self._h, C.cast(buf, C.c_void_p), int(rank), int(world)))
        self.comm_world = int(world)

    def comm_share(self, src):
        """Local: use the communicator of `src` (another context of this rank on the same GPU) -- one communicator per rank,
        whatever the number of batches in flight; the library orders the all-gathers issued through it."""
        _abi.check(self.lib.mocap_comm_share(self._h, src._h))
        self.comm_world = src.comm_world

    def comm_destroy(self):
        _abi.check(self.lib.mocap_comm_destroy(self._h))
        self.comm_world = 1

    def allgather_centroids(self, local, out=None):
        """local: this rank's centroid records (int32, contiguous, on this GPU) -> [world * len(local), ...] on every
        rank, in rank order; asynchronous on the current stream."""
        assert local.is_cuda and local.is_contiguous() and local.dtype == torch.int32
        world = self.comm_world
        if out is None:
            out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=torch.int32, device=local.device)
        assert out.is_contiguous() and out.numel() == world * local.numel() and out.dtype == torch.int32
        _abi.check(self.lib.mocap_allgather_centroids(self._h, _ptr(local), _ptr(out), local.numel(), _stream()))
        return out

    # ---- profiling -----------------------------------------------------------------------------------------------
    def tile_stats(self):
        """(tiles, tiles resolved by the dark-tile early-out) of the most recent blob_centroids batch"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        _abi.check(self.lib.mocap_tile_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def profile(self, on=True):
        _abi.check(self.lib.mocap_profile_enable(self._h, int(on)))

    def profile_read(self):
        """Accumulated HIP-event milliseconds / launch counts per kernel since the last read (mocap_hip.h)."""
        ms = (C.c_double * 5)()
        n = (C.c_int * 5)()
        _abi.check(self.lib.mocap_profile_read(self._h, ms, n))
        return {"filter_ms": ms[0], "filter_launches": n[0], "contour_ms": ms[1], "contour_launches": n[1],
                "corr_ms": ms[2], "corr_launches": n[2], "scan_ms": ms[3], "scan_launches": n[3],
                "settle_ms": ms[4], "settle_launches": n[4]}


class BAProblem:
    """The residual function of the reference's bundle adjustment (lib/Helpers.py:161-167) with everything that does not
    change between evaluations kept on the GPU: image points [N, C, 2] and validity [N, C] are uploaded once, the
    intrinsics live in the context's camera table (set_cameras), and an evaluation is one C-ABI call (mocap_ba_residuals:
    one launch -- rotvec -> R, triangulation, reprojection, float32 cast on the device -- and one stream wait)."""

    def __init__(self, ctx, pts, valid=None):
        pts = np.ascontiguousarray(pts, np.float64)
        assert pts.ndim == 3 and pts.shape[2] == 2, pts.shape
        self.ctx, (self.N, self.C) = ctx, pts.shape[:2]
        valid = np.ones((self.N, self.C), np.uint8) if valid is None else np.ascontiguousarray(valid, np.uint8).reshape(self.N, self.C)
        self.d_pts = torch.from_numpy(pts).to(ctx.device)
        self.d_valid = torch.from_numpy(valid).to(ctx.device)
        self._res = np.empty((1, self.N), np.float32)
        self._cnt = np.empty(1, np.int32)

    def residuals(self, params):
        """params: one parameter vector [6 (C - 1)] -> float32 residual vector, or a batch [B, 6 (C - 1)] -> list of
        residual vectors (a vector is shorter than N only when groups hold [None, None] entries, as in the reference)."""
        p = np.ascontiguousarray(params, np.float64)
        single = p.ndim == 1
        p = p.reshape(-1, 6 * (self.C - 1))
        B = p.shape[0]
        if self._res.shape[0] < B:
            self._res = np.empty((B, self.N), np.float32)
            self._cnt = np.empty(B, np.int32)
        ctx = self.ctx
        _abi.check(ctx.lib.mocap_ba_residuals(ctx._h, p.ctypes.data_as(C.POINTER(C.c_double)), B, _ptr(self.d_pts), _ptr(self.d_valid),
                                              self.N, self.C, self._res.ctypes.data_as(C.POINTER(C.c_float)),
                                              self._cnt.ctypes.data_as(C.POINTER(C.c_int)), _stream()))
        if single:
            return self._res[0, :self._cnt[0]].copy()
        return [self._res[b, :self._cnt[b]].copy() for b in range(B)]


def comm_available():
    """Local check, no communication: can this process load RCCL (mocap_comm_available)?  Raises MocapError when not."""
    _abi.check(_abi.load().mocap_comm_available())
    return True


def comm_unique_id():
    """MOCAP_COMM_ID_BYTES opaque bytes (ncclGetUniqueId) for MocapContext.comm_init; call on rank 0 only."""
    buf = (C.c_char * _abi.COMM_ID_BYTES)()
    _abi.check(_abi.load().mocap_comm_unique_id(C.cast(buf, C.c_void_p)))
    return bytes(buf)


_contexts = threading.local()


def default_context(width=1, height=1, n_slots=1, device=None):
    """Per-thread context cache keyed by geometry (the drop-in modules under mocapv2_amd.lib use it).  A context owns
    per-batch scratch, so its calls must not interleave; the reference runs one `_find_dot` loop per camera thread
    (RealtimeTracking_FLIR.py:307-312) -- each of those threads gets a context of its own here, and its buffers go
    with the thread."""
    if device is None:
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    key = (int(width), int(height), int(n_slots), int(device))
    cache = _contexts.__dict__.setdefault("cache", {})
    ctx = cache.get(key)
    if ctx is None:
        ctx = cache[key] = MocapContext(width, height, n_slots, device)
    return ctx
