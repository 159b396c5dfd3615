"""Throughput of the Bayer -> gray kernel on a bench-sized batch (3072 x 1080p).  python scratch/bayer_gray_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocapv2_amd.engine import MocapContext  # noqa: E402

ctx = MocapContext(8, 8)
n, H, W = 3072, 1080, 1920
raw = torch.randint(0, 256, (n, H, W), dtype=torch.uint8, device="cuda")
out = torch.empty_like(raw)
for _ in range(3):
    ctx.bayer_gray(raw, 3, 14, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
e0.record()
for _ in range(reps):
    ctx.bayer_gray(raw, 3, 14, out=out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
gb = 2 * n * H * W / 1e9
print(f"bayer_gray_kernel: {ms:.3f} ms per {n} images, {gb / ms:.2f} TB/s read+write ({n * H * W / 1e9 / ms:.2f} TB/s of frame bytes)")
