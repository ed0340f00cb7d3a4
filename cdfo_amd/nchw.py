"""Host wrappers of the small NCHW operators (csrc/nchw_ops.hip) used by the DCN consumer modules."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import check
from .kernels import _stream, ACT_NONE


def _p(t):
    return C.c_void_p(None if t is None else t.data_ptr())


def _need(t: torch.Tensor):
    if not t.is_cuda:
        raise NotImplementedError("HIP path: CPU tensors are not supported")
    return t.contiguous().float()


def conv2d(x, weight, bias=None, stride=1, pad=0, act=ACT_NONE):
    x, weight = _need(x), _need(weight)
    B, Cc, H, W = x.shape
    Co, Ci, kh, kw = weight.shape
    assert Ci == Cc
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    out = torch.empty((B, Co, Ho, Wo), dtype=torch.float32, device=x.device)
    check(_lib.lib().cdfo_conv2d_nchw(_p(x), _p(weight), _p(None if bias is None else _need(bias)), B, Cc, H, W, Co, kh,
                                      kw, stride, pad, act, _p(out), _stream()), "cdfo_conv2d_nchw")
    return out


def maxpool(x, k, stride):
    x = _need(x)
    B, Cc, H, W = x.shape
    out = torch.empty((B, Cc, (H - k) // stride + 1, (W - k) // stride + 1), dtype=torch.float32, device=x.device)
    check(_lib.lib().cdfo_maxpool_nchw(_p(x), B * Cc, H, W, k, stride, _p(out), _stream()), "cdfo_maxpool_nchw")
    return out


def resize_bilinear(x, Ho, Wo, out: Optional[torch.Tensor] = None, accumulate=False):
    x = _need(x)
    B, Cc, H, W = x.shape
    if out is None:
        out = torch.empty((B, Cc, Ho, Wo), dtype=torch.float32, device=x.device)
    check(_lib.lib().cdfo_resize_bilinear_nchw(_p(x), B * Cc, H, W, Ho, Wo, int(accumulate), _p(out), _stream()),
          "cdfo_resize_bilinear_nchw")
    return out


def avgpool(x):
    x = _need(x)
    B, Cc, H, W = x.shape
    out = torch.empty((B, Cc, 1, 1), dtype=torch.float32, device=x.device)
    check(_lib.lib().cdfo_avgpool_nchw(_p(x), B * Cc, C.c_longlong(H * W), _p(out), _stream()), "cdfo_avgpool_nchw")
    return out


def ew(a, mode, b=None, x=None, y=None):
    a = _need(a)
    out = torch.empty_like(a)
    P = a.shape[-1] * a.shape[-2]
    check(_lib.lib().cdfo_ew_nchw(_p(a), _p(None if b is None else _need(b)), _p(None if x is None else _need(x)),
                                  _p(None if y is None else _need(y)), C.c_longlong(a.numel()), C.c_longlong(P), mode,
                                  _p(out), _stream()), "cdfo_ew_nchw")
    return out
