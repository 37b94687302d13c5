"""Device-side image preprocessing.

The reference hands lists of PIL images to HF AutoImageProcessor on the CPU every step (swin.py:131; for
microsoft/swin-tiny-patch4-window7-224 that is a PIL BICUBIC resize to 224x224 on uint8, x 1/255, ImageNet mean/std).  Here decoded
images are uint8 HWC tensors (any size) already on the device; two launches (horizontal, vertical pass) resize with Pillow's own
fixed-point arithmetic, rescale, normalise and lay the batch out as the bf16 [B,3,H,W] tensor the patch-embedding kernels read.
"""
import ctypes as _c
from typing import List, Sequence

import torch

from . import ops
from ._lib import load_library

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


_COEF_CACHE = {}


def _cubic(x: float) -> float:
    """Keys cubic a = -0.5, the kernel of Pillow's BICUBIC filter (support 2)."""
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def bicubic_coeffs(in_size: int, out_size: int):
    """Per-output-coordinate source window [xmin, xmin + n) and 22-bit fixed-point weights of Pillow's antialiased BICUBIC resampler
    (what the reference's HF image processor calls, swin.py:131).  Plain-Python double arithmetic in the order Pillow's C code
    uses, so the integer weights are the same.  Returns (bounds int32 [out, 2], weights int32 [out, ksize], ksize)."""
    key = (in_size, out_size)
    if key in _COEF_CACHE:
        return _COEF_CACHE[key]
    import math

    import numpy as np
    scale = in_size / out_size
    fscale = scale if scale >= 1.0 else 1.0
    support = 2.0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / fscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = []
        ww = 0.0
        for x in range(xmax):
            v = _cubic((x + xmin - center + 0.5) * ss)
            w.append(v)
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << 22)) if k < 0 else int(0.5 + k * (1 << 22))
        bounds[xx] = (xmin, xmax)
    _COEF_CACHE[key] = (bounds, kk, ksize)
    return _COEF_CACHE[key]


def preprocess_images_bicubic(images: Sequence[torch.Tensor], size: int = 224, mean=IMAGENET_MEAN, std=IMAGENET_STD,
                              rescale: float = 1.0 / 255.0, out: torch.Tensor = None, return_uint8: bool = False):
    """The reference's preprocessing (HF image processor: PIL BICUBIC resize to size x size on uint8, x 1/255, (x - mean) / std),
    on the device.  images: B uint8 tensors [H_b, W_b, 3] on the GPU (contiguous).  Returns bf16 [B, 3, size, size]
    (and the resized uint8 [B, size, size, 3] when return_uint8)."""
    import numpy as np
    lib = load_library()
    B = len(images)
    if B == 0:
        raise ValueError("preprocess_images_bicubic: empty batch")
    for im in images:
        if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3 or not im.is_contiguous():
            raise TypeError("preprocess_images_bicubic: every image must be a contiguous uint8 [H, W, 3] tensor")
        ops._require_gpu(im, "medmoe_preprocess_bicubic")
    dev = images[0].device
    coef, offs, meta, row0 = [], {}, [], 0

    def table(n_in):
        if n_in not in offs:
            b, k, ks = bicubic_coeffs(n_in, size)
            pos = sum(c.size for c in coef)
            coef.append(b.reshape(-1)); coef.append(k.reshape(-1))
            offs[n_in] = (pos, pos + b.size, ks)
        return offs[n_in]
    for im in images:
        H, W = int(im.shape[0]), int(im.shape[1])
        bh, kh, ksh = table(W)
        bv, kv, ksv = table(H)
        meta.append([H, W, ksh, ksv, bh, kh, bv, kv, row0])
        row0 += H
    coef_t = torch.from_numpy(np.concatenate(coef).astype(np.int32)).to(dev, non_blocking=True)
    meta_t = torch.tensor(meta, dtype=torch.int64).to(dev, non_blocking=True)
    ptrs = torch.tensor([im.data_ptr() for im in images], dtype=torch.int64).to(dev, non_blocking=True)
    tmp = torch.empty(row0 * size * 3, device=dev, dtype=torch.uint8)
    if out is None:
        out = torch.empty(B, 3, size, size, device=dev, dtype=torch.bfloat16)
    u8 = torch.empty(B, size, size, 3, device=dev, dtype=torch.uint8) if return_uint8 else None
    m = (_c.c_float * 3)(*mean)
    s_ = (_c.c_float * 3)(*std)
    rc = lib.medmoe_preprocess_bicubic(_c.c_void_p(ptrs.data_ptr()), _c.c_void_p(meta_t.data_ptr()), _c.c_void_p(coef_t.data_ptr()),
                                       _c.c_void_p(tmp.data_ptr()), _c.c_void_p(out.data_ptr()), _c.c_void_p(u8.data_ptr() if u8 is not None else 0),
                                       _c.c_int(B), _c.c_int(size), _c.c_int(size), _c.c_longlong(row0), _c.c_float(rescale), m, s_, ops._stream())
    ops._chk(rc, "preprocess_bicubic")
    return (out, u8) if return_uint8 else out


def preprocess_images(images: Sequence[torch.Tensor], size: int = 224, mean=IMAGENET_MEAN, std=IMAGENET_STD,
                      rescale: float = 1.0 / 255.0, out: torch.Tensor = None, resample: str = "bicubic") -> torch.Tensor:
    """images: B uint8 tensors [H_b, W_b, 3] on the GPU (contiguous).  Returns bf16 [B, 3, size, size].
    resample "bicubic" (default) = the reference's filter (preprocess_images_bicubic); "bilinear" = torch's half-pixel bilinear."""
    if resample == "bicubic":
        return preprocess_images_bicubic(images, size, mean, std, rescale, out)
    if resample != "bilinear":
        raise ValueError("resample must be 'bicubic' or 'bilinear'")
    lib = load_library()
    B = len(images)
    if B == 0:
        raise ValueError("preprocess_images: empty batch")
    for im in images:
        if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3 or not im.is_contiguous():
            raise TypeError("preprocess_images: every image must be a contiguous uint8 [H, W, 3] tensor")
        ops._require_gpu(im, "medmoe_preprocess")
    dev = images[0].device
    ptrs = torch.tensor([im.data_ptr() for im in images], dtype=torch.int64).to(dev, non_blocking=True)
    hw = torch.tensor([[im.shape[0], im.shape[1]] for im in images], dtype=torch.int32).to(dev, non_blocking=True)
    if out is None:
        out = torch.empty(B, 3, size, size, device=dev, dtype=torch.bfloat16)
    m = (_c.c_float * 3)(*mean)
    s = (_c.c_float * 3)(*std)
    rc = lib.medmoe_preprocess(_c.c_void_p(ptrs.data_ptr()), _c.c_void_p(hw.data_ptr()), _c.c_void_p(out.data_ptr()), _c.c_int(B),
                               _c.c_int(size), _c.c_int(size), _c.c_float(rescale), m, s, ops._stream())
    ops._chk(rc, "preprocess")
    return out
