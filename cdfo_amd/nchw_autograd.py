"""torch.autograd Functions over the small NCHW operators (csrc/nchw_ops.hip forward, csrc/nchw_bwd.hip backward): what the
two consumer modules of the deformable convolution need to be TRAINABLE like their reference classes -- ``DSTA``
(ops/attentionlayer.py:117-156) entirely, ``MVDualAttAlignment`` (arch/SIDECVSR_our.py:3303-3352) for its layout changes and
its offset / mask assembly (its convolutions and channel attention are the pixel-major Functions of ``cdfo_amd/autograd.py``).

Same rule as the CVSR_V8 training path: every operator that touches pixels runs in libcdfo_hip.so in both directions; torch
supplies the graph, device memory and arithmetic on parameter-sized tensors only.  No CPU fallback."""
from __future__ import annotations

import ctypes as C

import torch
from torch.autograd import Function

from . import _lib
from . import kernels as K
from . import nchw as N
from ._lib import check
from .kernels import ACT_NONE, ACT_RELU, ACT_SIGMOID, _stream


def _p(t):
    return C.c_void_p(None if t is None else t.data_ptr())


def _c(t):
    return t.detach().contiguous().float()


def _ew_bwd(mode, g, y=None, a=None, x=None, yv=None, P=1, shape=None):
    out = torch.empty(shape if shape is not None else g.shape, dtype=torch.float32, device=g.device)
    check(_lib.lib().cdfo_ew_nchw_bwd(_p(g), _p(y), _p(a), _p(x), _p(yv), C.c_longlong(out.numel()), C.c_longlong(P), mode, _p(out),
                                      _stream()), "cdfo_ew_nchw_bwd")
    return out


class _Conv2d(Function):
    """y = act(conv2d(x, w, b)), any stride / padding, act in {none, relu, sigmoid}."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, act):
        x, w = _c(x), _c(weight)
        y = N.conv2d(x, w, None if bias is None else _c(bias), stride, pad, act)
        ctx.meta = (stride, pad, act, bias is not None)
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, g):
        stride, pad, act, has_bias = ctx.meta
        x, w, y = ctx.saved_tensors
        g = _c(g)
        if act == ACT_RELU:
            g = _ew_bwd(1, g, y=y)
        elif act == ACT_SIGMOID:
            g = _ew_bwd(2, g, y=y)
        B, Cc, H, W = x.shape
        Co, _, kh, kw = w.shape
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], has_bias and ctx.needs_input_grad[2]
        gx = torch.empty_like(x) if need_x else None
        gw = torch.empty_like(w) if (need_w or need_b) else None
        gb = torch.empty(Co, dtype=torch.float32, device=x.device) if need_b else None
        check(_lib.lib().cdfo_conv2d_nchw_bwd(_p(x), _p(w), _p(g), B, Cc, H, W, Co, kh, kw, stride, pad, _p(gx), _p(gw), _p(gb),
                                              _stream()), "cdfo_conv2d_nchw_bwd")
        return gx, (gw if need_w else None), gb, None, None, None


class _MaxPool(Function):
    @staticmethod
    def forward(ctx, x, k, stride):
        x = _c(x)
        ctx.meta = (k, stride)
        ctx.save_for_backward(x)
        return N.maxpool(x, k, stride)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        k, stride = ctx.meta
        g = _c(g)
        B, Cc, H, W = x.shape
        idx = torch.empty(g.numel(), dtype=torch.int32, device=x.device)
        gx = torch.empty_like(x)
        check(_lib.lib().cdfo_maxpool_nchw_bwd(_p(x), _p(g), B * Cc, H, W, k, stride, _p(idx), _p(gx), _stream()),
              "cdfo_maxpool_nchw_bwd")
        return gx, None, None


class _Resize(Function):
    """F.interpolate(x, size=(Ho, Wo), mode='bilinear', align_corners=False)"""

    @staticmethod
    def forward(ctx, x, Ho, Wo):
        x = _c(x)
        ctx.meta = (tuple(x.shape), Ho, Wo)
        return N.resize_bilinear(x, Ho, Wo)

    @staticmethod
    def backward(ctx, g):
        (B, Cc, H, W), Ho, Wo = ctx.meta
        g = _c(g)
        gx = torch.empty((B, Cc, H, W), dtype=torch.float32, device=g.device)
        check(_lib.lib().cdfo_resize_bilinear_nchw_bwd(_p(g), B * Cc, H, W, Ho, Wo, _p(gx), _stream()), "cdfo_resize_bilinear_nchw_bwd")
        return gx, None, None


class _PlaneMean(Function):
    """adaptive_avg_pool2d(x, 1) -> [B, C, 1, 1]"""

    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        ctx.shape = tuple(x.shape)
        return N.avgpool(x)

    @staticmethod
    def backward(ctx, g):
        B, Cc, H, W = ctx.shape
        return _ew_bwd(4, _c(g), P=H * W, shape=ctx.shape)


class _Unary(Function):
    """relu (mode 1) / sigmoid (mode 2)"""

    @staticmethod
    def forward(ctx, x, mode):
        y = N.ew(_c(x), mode)
        ctx.mode = mode
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return _ew_bwd(ctx.mode, _c(g), y=y), None


class _Add(Function):
    @staticmethod
    def forward(ctx, a, b):
        return N.ew(_c(a), 0, b=_c(b))

    @staticmethod
    def backward(ctx, g):
        return g, g


class _Gate(Function):
    """DSTA's output: x * sigmoid(a) * yv[b][c] (ops/attentionlayer.py:154-156), yv = the channel gate [B, C, 1, 1]."""

    @staticmethod
    def forward(ctx, a, x, yv):
        a, x, yv = _c(a), _c(x), _c(yv)
        ctx.save_for_backward(a, x, yv)
        return N.ew(a, 3, x=x, y=yv)

    @staticmethod
    def backward(ctx, g):
        a, x, yv = ctx.saved_tensors
        g = _c(g)
        B, Cc, H, W = x.shape
        P = H * W
        ga = _ew_bwd(5, g, a=a, x=x, yv=yv, P=P)
        gx = _ew_bwd(6, g, a=a, yv=yv, P=P)
        gy = torch.empty((B, Cc, 1, 1), dtype=torch.float32, device=x.device)
        check(_lib.lib().cdfo_gate_nchw_bwd_y(_p(g), _p(a), _p(x), B * Cc, C.c_longlong(P), _p(gy), _stream()), "cdfo_gate_nchw_bwd_y")
        return ga, gx, gy


class _ToPixelMajor(Function):
    """[B, C, H, W] -> [B, H, W, C] (the layout of cdfo_amd/autograd.py's Functions); the two transposes are each other's adjoint"""

    @staticmethod
    def forward(ctx, x):
        return K.nchw_to_nhwc(_c(x))

    @staticmethod
    def backward(ctx, g):
        return K.nhwc_to_nchw(_c(g))


class _ToNCHW(Function):
    @staticmethod
    def forward(ctx, x):
        return K.nhwc_to_nchw(_c(x))

    @staticmethod
    def backward(ctx, g):
        return K.nchw_to_nhwc(_c(g))


class _OffsetMask(Function):
    """MVDualAttAlignment's offset / mask assembly (arch.py:3336-3350) from the two conv_offset outputs h1, h2 (pixel-major
    [B, H, W, 27 dg]) and the motion field (NCHW [B, 2, H, W], no gradient): offset [B, 18 dg, H, W], mask [B, 9 dg, H, W]."""

    @staticmethod
    def forward(ctx, h1, h2, flow, third, mag):
        h1, h2 = _c(h1), _c(h2)
        B, H, W, _ = h1.shape
        P = H * W
        offset = torch.empty((B, 2 * third, H, W), dtype=torch.float32, device=h1.device)
        mask = torch.empty((B, third, H, W), dtype=torch.float32, device=h1.device)
        check(_lib.lib().cdfo_mv_offset_mask(_p(h1), _p(h2), h1.stride(2), _p(flow), C.c_longlong(2 * P), B, C.c_longlong(P), third,
                                             float(mag), _p(offset), _p(mask), _stream()), "cdfo_mv_offset_mask")
        ctx.meta = (third, float(mag))
        ctx.save_for_backward(h1, h2)
        return offset, mask

    @staticmethod
    def backward(ctx, goff, gmask):
        h1, h2 = ctx.saved_tensors
        third, mag = ctx.meta
        B, H, W, _ = h1.shape
        g1, g2 = torch.empty_like(h1), torch.empty_like(h2)
        check(_lib.lib().cdfo_mv_offset_mask_bwd(_p(h1), _p(h2), h1.stride(2), _p(_c(goff)), _p(_c(gmask)), B, C.c_longlong(H * W),
                                                 third, mag, _p(g1), _p(g2), _stream()), "cdfo_mv_offset_mask_bwd")
        return g1, g2, None, None, None


def conv2d(x, weight, bias=None, stride=1, pad=0, act=ACT_NONE):
    return _Conv2d.apply(x, weight, bias, stride, pad, act)


def maxpool(x, k, stride):
    return _MaxPool.apply(x, k, stride)


def resize_bilinear(x, Ho, Wo):
    return _Resize.apply(x, Ho, Wo)


plane_mean = _PlaneMean.apply
add = _Add.apply
gate = _Gate.apply
to_pixel_major = _ToPixelMajor.apply
to_nchw = _ToNCHW.apply


def relu(x):
    return _Unary.apply(x, 1)


def sigmoid(x):
    return _Unary.apply(x, 2)


def offset_mask(h1, h2, flow, third, mag):
    return _OffsetMask.apply(h1, h2, flow, third, mag)
