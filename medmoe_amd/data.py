"""Device-side image preprocessing.

The reference hands lists of PIL images to HF AutoImageProcessor on the CPU every step (swin.py:131; for
microsoft/swin-tiny-patch4-window7-224 that is resize to 224x224, x 1/255, ImageNet mean/std).  Here decoded images
are uint8 HWC tensors (any size) already on the device; one launch resizes, rescales, normalises and lays the batch out
as the bf16 [B,3,H,W] tensor the patch-embedding kernels read.
"""
import ctypes as _c
from typing import List, Sequence

import torch

from . import ops
from ._lib import load_library

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def preprocess_images(images: Sequence[torch.Tensor], size: int = 224, mean=IMAGENET_MEAN, std=IMAGENET_STD,
                      rescale: float = 1.0 / 255.0, out: torch.Tensor = None) -> torch.Tensor:
    """images: B uint8 tensors [H_b, W_b, 3] on the GPU (contiguous).  Returns bf16 [B, 3, size, size]."""
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
