"""PSNR / SSIM of single-channel frames on the device, with the reference's semantics (metric/psnr_ssim.py:278-317,
320-399, and the per-frame loop of cal_psnr_ssim at :446-484): 0..255 scale, ``crop_border`` pixels dropped, fp64 sums.

The reference saves the network output as an 8-bit PNG, reads it back and compares with the ground-truth PNG; here the
tensors stay in HBM.  ``from_unit_range=True`` multiplies by 255 and clamps (the network emits [0,1] values);
``round8=True`` additionally rounds to integers, which is what the PNG round trip does.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib
from .kernels import _stream, _vp, on_device


def _partials(a: torch.Tensor, b: torch.Tensor, crop: int, metric: int, from_unit_range: bool, round8: bool):
    if not (a.is_cuda and b.is_cuda):
        raise NotImplementedError("cdfo_amd.metrics: device tensors only (HIP path, no CPU fallback)")
    if a.shape != b.shape:
        raise ValueError(f"Image shapes are different: {tuple(a.shape)}, {tuple(b.shape)}.")
    H, W = a.shape[-2], a.shape[-1]
    a = a.reshape(-1, H, W).contiguous().float()
    b = b.reshape(-1, H, W).contiguous().float()
    N = a.shape[0]
    part = torch.empty((N, 1024), dtype=torch.float64, device=a.device)
    nb = C.c_int(0)
    with on_device(a):
        _lib.check(_lib.lib().cdfo_metric_partials(_vp(a), _vp(b), N, H, W, crop,
                                                   C.c_float(255.0 if from_unit_range else 1.0), int(from_unit_range),
                                                   int(round8), metric, _vp(part), part.numel(), C.byref(nb), _stream()),
                   "cdfo_metric_partials")
    # the kernel packs its partial sums as [N][nblocks]; fixed-order fp64 sum per frame
    return part.view(-1)[:N * nb.value].view(N, nb.value).sum(dim=1), H - 2 * crop, W - 2 * crop


def calculate_psnr(img1: torch.Tensor, img2: torch.Tensor, crop_border: int = 4, from_unit_range: bool = True,
                   round8: bool = False) -> torch.Tensor:
    """Per-frame PSNR (fp64 tensor [N]); ``inf`` where the frames are identical."""
    s, Hc, Wc = _partials(img1, img2, crop_border, 0, from_unit_range, round8)
    mse = s / float(Hc * Wc)
    return torch.where(mse == 0, torch.full_like(mse, float("inf")), 20.0 * torch.log10(255.0 / torch.sqrt(mse)))


def calculate_ssim(img1: torch.Tensor, img2: torch.Tensor, crop_border: int = 4, from_unit_range: bool = True,
                   round8: bool = False) -> torch.Tensor:
    """Per-frame SSIM (fp64 tensor [N]): mean of the 11x11-Gaussian SSIM map over its valid positions."""
    s, Hc, Wc = _partials(img1, img2, crop_border, 1, from_unit_range, round8)
    return s / float((Hc - 10) * (Wc - 10))
