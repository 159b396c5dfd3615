"""Mirror of reference lib/ImageOperations.py: `_find_dot`, `image_filter_gpu`, `image_filter_cpu`, the module
globals `cuda_lock`, `camera_params`, `intrinsics_json`.

Differences a caller can see (all display-only): the returned image carries no drawContours/putText/circle
overlay (reference :52-55,67-73) -- it is the undistorted frame (or the filtered mask with return_filtered=True).
The intrinsics file is read on first use instead of at import, so importing from another working directory works;
a missing file raises the same FileNotFoundError, just later.
"""
import json
import threading

import numpy as np
import torch

from ..engine import default_context

cuda_lock = threading.Lock()  # kept for callers that take it; the HIP path needs no global lock

intrinsics_json = "./jsons/camera-params-in.json"
camera_params = None


def _params():
    global camera_params
    if camera_params is None:
        with open(intrinsics_json) as f:
            camera_params = json.load(f)
    return camera_params


def _ctx_for(image):
    h, w = image.shape
    return default_context(w, h, 1)


def _upload(ctx, image):
    return torch.from_numpy(np.ascontiguousarray(image, dtype=np.uint8)).to(ctx.device)


def image_filter_cpu(image, camera_number=0):
    """reference :15-21 -- 5x5 median, then threshold 0.85*255 (no box blur)."""
    ctx = _ctx_for(image)
    return ctx.image_filter(_upload(ctx, image), order=1).cpu().numpy()


def image_filter_gpu(image, camera_number=0):
    """reference :23-31 -- 5x5 in-bounds box mean, threshold 0.85*255, 5x5 median."""
    ctx = _ctx_for(image)
    return ctx.image_filter(_upload(ctx, image), order=0).cpu().numpy()


def _find_dot(img, print_location=False, return_filtered=False):
    """reference :33-78 -- (image, image_points); image_points = [[cx, cy], ...] in cv.findContours order or
    [[None, None]] when nothing passes the area/circularity gate.  Camera 0's intrinsics are used for every
    camera, as in the reference (:37)."""
    assert img.ndim == 2, "expects a single-channel image"
    params = _params()
    camera_number = 0
    ctx = _ctx_for(img)
    ctx.set_undistort(0, np.array(params[camera_number]["intrinsic_matrix"]), np.array(params[camera_number]["distortion_coef"]))
    d_img = _upload(ctx, img)
    xy, cnt = ctx.record_views(ctx.blob_centroids(d_img, cam_mod=1))
    out_img = ctx.image_filter(d_img, order=0, slot=0) if return_filtered else ctx.undistort(d_img, 0)
    n = int(cnt.cpu()[0])
    if n < 0:
        raise RuntimeError(f"blob kernel capacity exceeded (status {n}); see include/mocap_hip.h MOCAP_BLOB_E_*")
    if n > xy.shape[1]:
        raise RuntimeError(f"{n} image points exceed the record capacity {xy.shape[1]}")
    image_points = [[int(x), int(y)] for x, y in xy[0, :n].cpu().numpy()]
    if len(image_points) == 0:
        image_points = [[None, None]]
    return out_img.cpu().numpy(), image_points
