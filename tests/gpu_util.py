"""Helpers shared by the -m gpu parity tests."""
import numpy as np
import torch


def pack_mask(mask):
    """{0,nonzero}[n,H,W] -> int32 [n,H,ceil(W/32)] device tensor, bit b of word k = pixel 32k+b."""
    mask = np.asarray(mask)
    if mask.ndim == 2:
        mask = mask[None]
    n, H, W = mask.shape
    wpr = (W + 31) // 32
    bits = np.zeros((n, H, wpr * 32), np.uint8)
    bits[:, :, :W] = mask != 0
    packed = np.packbits(bits, axis=2, bitorder="little").reshape(n, H, wpr, 4)
    words = np.ascontiguousarray(packed).view(np.uint32).reshape(n, H, wpr)
    return torch.from_numpy(words.view(np.int32).copy()).cuda()


def unpack_mask(words, W):
    """int32 [n,H,wpr] tensor -> (bool [n,H,W], padding bits)"""
    w = np.ascontiguousarray(words.cpu().numpy()).view(np.uint32)
    n, H, wpr = w.shape
    b = np.unpackbits(w.view(np.uint8).reshape(n, H, wpr * 4), axis=2, bitorder="little")
    return b[:, :, :W].astype(bool), b[:, :, W:]
