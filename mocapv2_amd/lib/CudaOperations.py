"""Mirror of reference lib/CudaOperations.py: `fast_cuda_blur`, `fast_cuda_demosaic` on host NumPy arrays.

The reference JIT-compiles two numba.cuda kernels and pays two host->device copies plus one device->host copy per
call (lib/CudaOperations.py:28-41); here the image is uploaded once, a HIP kernel runs, the result comes back.
"""
import numpy as np
import torch

from ..engine import default_context


def _upload(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint8)).to(default_context().device)


def fast_cuda_blur(image: np.ndarray, kernel_size: int = 5) -> np.ndarray:
    """reference lib/CudaOperations.py:24-41 -- mean over the in-bounds taps, truncated to uint8."""
    assert len(image.shape) == 2, "Only grayscale images supported"
    ctx = default_context()
    return ctx.box_blur(_upload(image), kernel_size).cpu().numpy()


def fast_cuda_demosaic(bayer_img: np.ndarray) -> np.ndarray:
    """reference lib/CudaOperations.py:84-100 -- bilinear Bayer demosaic, output channels B,G,R."""
    assert len(bayer_img.shape) == 2, "Expected a single-channel Bayer image (grayscale)"
    ctx = default_context()
    return ctx.demosaic(_upload(bayer_img)).cpu().numpy()
