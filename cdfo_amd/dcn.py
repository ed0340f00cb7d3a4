"""Deformable convolution (DCNv1 / DCNv2) on libcdfo_hip.so -- boundary B2 of SURVEY section 8b.

Design of this file (it is NOT a transcription of the reference's wrapper): the reference keeps two autograd Functions,
each pre-allocating an output, handing two empty scratch tensors to a pybind module and tracking ``im2col_step``
(ops/dcn/deform_conv.py:14-172).  The HIP operator is one fused kernel family -- sampling and contraction in one launch,
no ``columns`` buffer, any batch size per launch -- so here there is

* one value type, :class:`Geometry`, that owns every shape rule of the operator (output size of
  ops/dcn/src/deform_conv_cuda.cpp:513-516, operand shapes of :506-511);
* one launcher pair, :func:`launch_forward` / :func:`launch_backward`, that validates operands and calls the C-ABI
  (``cdfo_dcn_forward_dt`` / ``cdfo_dcn_backward_dt`` of include/cdfo_hip.h) with a workspace that is cached per
  (device, stream) and only ever grows -- the pybind-shaped module ``cdfo_amd.deform_conv_cuda`` is a five-function shim
  over the same two launchers, for callers that bring the reference's own ``deform_conv.py``;
* ONE ``torch.autograd.Function`` (:class:`_DeformableConv`), parameterised by ``mask is None`` (DCNv1) and
  ``bias is None``; its backward asks the library only for the gradients autograd wants (NULL pointers skip work);
* the eight public names of ``ops/dcn/deform_conv.py`` as thin adapters with the reference's call signatures
  (:186-187, :190-201, :234, :264-275, :305, :311-337) and parameter names (``weight``, ``bias``, ``conv_offset``,
  ``conv_offset_mask``: the state_dict schema a drop-in must keep).

Error behaviour kept from the reference: CPU tensors raise ``NotImplementedError`` (deform_conv.py:46-47, 136-137),
a non-4D input ``ValueError`` (:26-29), an empty output ``ValueError`` (:182-183), non-contiguous input / weight
``RuntimeError`` at the extension boundary only (cpp:493-494) -- the functional entry points here make them contiguous.
``im2col_step`` is accepted and ignored: the fused kernel has no column buffer to step over.
"""
from __future__ import annotations

import ctypes as C
import threading
import math
import os
from dataclasses import dataclass

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib
from .kernels import _stream, on_device

# Forward arithmetic of the float operator.  False (default): shapes the fused fast kernels cover run on split-fp16 MFMA
# (fp32-grade: <= 2e-5 * max|out| against the C oracle; range-safe -- a sampled value outside fp16's range makes the
# library re-run the exact kernel).  True (or CDFO_DCN_EXACT=1): always the exact-fp32 MFMA kernel.
EXACT_FP32 = os.environ.get("CDFO_DCN_EXACT", "0") not in ("", "0")

_DTYPE_TAG = {torch.float32: 0, torch.float16: 1, torch.float64: 2}     # CDFO_DTYPE_* of include/cdfo_hip.h


def _two(v):
    """int | (int, int) -> (int, int)"""
    if isinstance(v, (tuple, list)):
        if len(v) != 2:
            raise ValueError(f"expected an int or a pair, got {v!r}")
        return int(v[0]), int(v[1])
    return int(v), int(v)


@dataclass(frozen=True)
class Geometry:
    """Everything but the tensors: kernel extent, stride, padding, dilation (h, w each), conv groups, deformable groups."""
    kh: int
    kw: int
    sh: int
    sw: int
    ph: int
    pw: int
    dh: int
    dw: int
    groups: int
    dg: int

    @classmethod
    def of(cls, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1) -> "Geometry":
        (sh, sw), (ph, pw), (dh, dw) = _two(stride), _two(padding), _two(dilation)
        return cls(int(weight.shape[2]), int(weight.shape[3]), sh, sw, ph, pw, dh, dw, int(groups), int(deformable_groups))

    @property
    def taps(self) -> int:
        return self.kh * self.kw

    def out_hw(self, H: int, W: int):
        """cpp:513-516"""
        return ((H + 2 * self.ph - (self.dh * (self.kh - 1) + 1)) // self.sh + 1,
                (W + 2 * self.pw - (self.dw * (self.kw - 1) + 1)) // self.sw + 1)

    def c_args(self):
        return (self.kh, self.kw, self.sh, self.sw, self.ph, self.pw, self.dh, self.dw, self.groups, self.dg)


class _Scratch:
    """Device scratch per (device index, stream handle).  Launches on one stream are ordered, so the next launch may
    overwrite what the previous one left; two streams never share a buffer.  Plumbing, no arithmetic.
    Only SMALL workspaces are cached (the forward's weight image and flags: a few MB): a request above ``CACHE_LIMIT`` -- the
    backward's per-tile column / grad_output copies, 2.7 GB at the alignment module's c3 shape -- is a plain allocation that torch's
    caching allocator recycles and ``torch.cuda.empty_cache()`` can give back.  The table is bounded (``MAX_ENTRIES`` streams, oldest
    evicted) and guarded by a lock: DCN calls may come from several threads / side streams."""
    CACHE_LIMIT = 64 << 20
    MAX_ENTRIES = 16
    _bufs: dict = {}
    _lock = threading.Lock()

    @classmethod
    def get(cls, device, nbytes: int):
        if nbytes > cls.CACHE_LIMIT or torch.cuda.is_current_stream_capturing():     # (a graph's private pool must not leak into the cache)
            return torch.empty(max(nbytes, 16), dtype=torch.uint8, device=device)
        key = (device.index, int(torch.cuda.current_stream(device).cuda_stream))
        with cls._lock:
            buf = cls._bufs.get(key)
            if buf is None or buf.numel() < nbytes:
                cls._bufs.pop(key, None)
                while len(cls._bufs) >= cls.MAX_ENTRIES:
                    cls._bufs.pop(next(iter(cls._bufs)))
                buf = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=device)
                cls._bufs[key] = buf
            return buf

    @classmethod
    def release(cls):
        with cls._lock:
            cls._bufs.clear()


def release_workspaces() -> None:
    """Drop the cached (small) operator workspaces; the large backward scratch is never cached."""
    _Scratch.release()


def _ptr(t):
    return C.c_void_p(None if t is None else t.data_ptr())


def _operand_dtype(*tensors) -> int:
    """Device tensors of ONE of the reference's three element types (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
    deform_conv_cuda_kernel.cu:258) -> the library's dtype tag."""
    dt = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise NotImplementedError("deformable convolution (HIP): CPU tensors are not supported")
        if t.dtype not in _DTYPE_TAG:
            raise RuntimeError(f'"deform_conv" not implemented for \'{t.dtype}\' (float, double and half are)')
        if dt is not None and t.dtype != dt:
            raise RuntimeError(f"expected scalar type {dt} but found {t.dtype}")
        dt = t.dtype
    return _DTYPE_TAG[dt]


def _check_shapes(x, offset, mask, weight, geo: Geometry):
    """cpp:506-511 and the offset / mask extents the kernels index (cu:583-631).  Returns (B, C, H, W, Co, Ho, Wo)."""
    if x.dim() != 4:
        raise ValueError(f"Expected 4D tensor as input, got {x.dim()}D tensor instead.")
    B, Cin, H, W = (int(v) for v in x.shape)
    Co, Ck, kh, kw = (int(v) for v in weight.shape)
    if (kh, kw) != (geo.kh, geo.kw):
        raise RuntimeError(f"Input shape and kernel shape wont match: ({geo.kh} x {geo.kw} vs {kh} x {kw}).")
    if Cin != Ck * geo.groups:
        raise RuntimeError(f"Input shape and kernel channels wont match: ({Cin} vs {Ck * geo.groups}).")
    Ho, Wo = geo.out_hw(H, W)
    if min(B, Co, Ho, Wo) <= 0:
        raise ValueError(f"convolution input is too small (output would be {B}x{Co}x{Ho}x{Wo})")
    want = (B, 2 * geo.dg * geo.taps, Ho, Wo)
    if tuple(offset.shape) != want:
        raise RuntimeError(f"invalid offset shape {tuple(offset.shape)}, expected {want}")
    if mask is not None and tuple(mask.shape) != (B, geo.dg * geo.taps, Ho, Wo):
        raise RuntimeError(f"invalid mask shape {tuple(mask.shape)}, expected {(B, geo.dg * geo.taps, Ho, Wo)}")
    return B, Cin, H, W, Co, Ho, Wo


def launch_forward(x, offset, mask, weight, bias, out, geo: Geometry) -> None:
    """``out[B,Co,Ho,Wo] = DCN(x; offset, mask, weight) + bias`` written into the caller's tensor.  ``mask is None`` = DCNv1.
    ``x`` and ``weight`` must be contiguous (cpp:493-494); the others are made so."""
    if not x.is_contiguous():
        raise RuntimeError("input tensor has to be contiguous")
    if not weight.is_contiguous():
        raise RuntimeError("weight tensor has to be contiguous")
    dt = _operand_dtype(x, weight, bias, offset, mask, out)
    B, Cin, H, W, Co, Ho, Wo = _check_shapes(x, offset, mask, weight, geo)
    if out.numel() != B * Co * Ho * Wo or not out.is_contiguous():
        raise RuntimeError("output must be a contiguous tensor of B*Co*Ho*Wo elements")
    offset = offset.contiguous()
    mask = None if mask is None else mask.contiguous()
    bias = None if bias is None else bias.contiguous()
    L = _lib.lib()
    with on_device(x):
        nbytes = int(L.cdfo_dcn_workspace_bytes_dt(dt, 0, B, Cin, H, W, Co, *geo.c_args()))
        if nbytes < 0:
            raise RuntimeError("deformable convolution (HIP): unsupported shape")
        if EXACT_FP32 and dt == 0:
            nbytes = x.numel() * 4      # room for the group-planar copy only: less than the fast kernels ask for
        ws = _Scratch.get(x.device, nbytes)
        _lib.check(L.cdfo_dcn_forward_dt(dt, _ptr(x), _ptr(offset), _ptr(mask), _ptr(weight), _ptr(bias), _ptr(out), B, Cin,
                                         H, W, Co, *geo.c_args(), _ptr(ws), C.c_longlong(nbytes), _stream()),
                   "cdfo_dcn_forward_dt")


def launch_backward(x, offset, mask, weight, grad_out, geo: Geometry, *, grad_x=None, grad_offset=None, grad_mask=None,
                    grad_weight=None, grad_bias=None, scale: float = 1.0) -> None:
    """Gradients of :func:`launch_forward` into whichever of the five tensors the caller passes (the others are skipped by
    the library).  ``grad_x`` / ``grad_weight`` / ``grad_bias`` are accumulated into, ``grad_offset`` / ``grad_mask`` are
    assigned (the extension's conventions, cpp:260-266, 373-378, 566-573); ``scale`` multiplies the weight gradient."""
    if not x.is_contiguous():
        raise RuntimeError("input tensor has to be contiguous")
    if not weight.is_contiguous():
        raise RuntimeError("weight tensor has to be contiguous")
    dt = _operand_dtype(x, offset, mask, weight, grad_out, grad_x, grad_offset, grad_mask, grad_weight, grad_bias)
    B, Cin, H, W, Co, Ho, Wo = _check_shapes(x, offset, mask, weight, geo)
    if tuple(grad_out.shape) != (B, Co, Ho, Wo):
        raise RuntimeError(f"invalid gradOutput shape {tuple(grad_out.shape)}, expected {(B, Co, Ho, Wo)}")
    for name, g, like in (("grad_input", grad_x, x), ("grad_offset", grad_offset, offset), ("grad_mask", grad_mask, mask),
                          ("grad_weight", grad_weight, weight)):
        if g is not None and (like is None or g.numel() != like.numel() or not g.is_contiguous()):
            raise RuntimeError(f"{name} must be a contiguous tensor shaped like its forward counterpart")
    if grad_bias is not None and (grad_bias.numel() != Co or not grad_bias.is_contiguous()):
        raise RuntimeError("grad_bias must be a contiguous tensor of Co elements")
    offset, grad_out = offset.contiguous(), grad_out.contiguous()
    mask = None if mask is None else mask.contiguous()
    L = _lib.lib()
    with on_device(x):
        nbytes = int(L.cdfo_dcn_workspace_bytes_dt(dt, 1, B, Cin, H, W, Co, *geo.c_args()))
        if nbytes < 0:
            raise RuntimeError("deformable convolution (HIP): unsupported shape")
        ws = _Scratch.get(x.device, nbytes)
        _lib.check(L.cdfo_dcn_backward_dt(dt, _ptr(x), _ptr(offset), _ptr(mask), _ptr(weight), _ptr(grad_out), _ptr(grad_x),
                                          _ptr(grad_offset), _ptr(grad_mask), _ptr(grad_weight), _ptr(grad_bias), B, Cin, H,
                                          W, Co, *geo.c_args(), float(scale), _ptr(ws), C.c_longlong(nbytes), _stream()),
                   "cdfo_dcn_backward_dt")


class _DeformableConv(Function):
    """y = DCN(x, offset[, mask]; weight[, bias]).  DCNv1 is the call with ``mask is None`` and ``bias is None``."""

    @staticmethod
    def forward(ctx, x, offset, mask, weight, bias, geo: Geometry):
        for t in (x, offset, mask, weight, bias):
            if t is not None and not t.is_cuda:
                raise NotImplementedError("deformable convolution (HIP): CPU tensors are not supported")
        x, weight, offset = x.contiguous(), weight.contiguous(), offset.contiguous()
        mask = None if mask is None else mask.contiguous()
        B, _, H, W = _check_shapes(x, offset, mask, weight, geo)[:4]
        out = x.new_empty((B, weight.shape[0], *geo.out_hw(H, W)))
        launch_forward(x, offset, mask, weight, bias, out, geo)
        ctx.geo = geo
        ctx.save_for_backward(x, offset, mask, weight)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        if not grad_out.is_cuda:
            raise NotImplementedError("deformable convolution (HIP): CPU tensors are not supported")
        x, offset, mask, weight = ctx.saved_tensors
        need_x, need_off, need_mask, need_w, need_b = ctx.needs_input_grad[:5]
        g = dict(
            grad_x=torch.zeros_like(x) if need_x else None,                             # accumulated into: zero-filled
            grad_offset=torch.empty_like(offset) if need_off else None,                 # assigned
            grad_mask=torch.empty_like(mask) if (need_mask and mask is not None) else None,
            grad_weight=torch.zeros_like(weight) if need_w else None,
            grad_bias=weight.new_zeros(weight.shape[0]) if need_b else None)
        if any(v is not None for v in g.values()):
            launch_backward(x, offset, mask, weight, grad_out, ctx.geo, **g)
        return g["grad_x"], g["grad_offset"], g["grad_mask"], g["grad_weight"], g["grad_bias"], None


def deform_conv(input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1, im2col_step=64):
    """DCNv1, signature of ``DeformConvFunction.apply`` (deform_conv.py:16-25, 186)."""
    if input is not None and input.dim() != 4:
        raise ValueError(f"Expected 4D tensor as input, got {input.dim()}D tensor instead.")
    return _DeformableConv.apply(input, offset, None, weight, None,
                                 Geometry.of(weight, stride, padding, dilation, groups, deformable_groups))


def modulated_deform_conv(input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
                          deformable_groups=1):
    """DCNv2, signature of ``ModulatedDeformConvFunction.apply`` (deform_conv.py:116-126, 187)."""
    return _DeformableConv.apply(input, offset, mask, weight, bias,
                                 Geometry.of(weight, stride, padding, dilation, groups, deformable_groups))


class DeformConvFunction:
    """Name kept for callers that spell ``DeformConvFunction.apply(...)`` (deform_conv.py:14)."""
    apply = staticmethod(deform_conv)


class ModulatedDeformConvFunction:
    """Name kept for callers that spell ``ModulatedDeformConvFunction.apply(...)`` (deform_conv.py:114)."""
    apply = staticmethod(modulated_deform_conv)


class _DeformableConvModule(nn.Module):
    """Shared parameter schema of the four module classes: ``weight [Co, C/groups, kh, kw]`` drawn uniformly in
    +-1/sqrt(C*kh*kw), an optional zero-initialised ``bias [Co]`` (deform_conv.py:214-220, 294-302)."""

    def _make_parameters(self, in_channels, out_channels, kernel_size, groups, deformable_groups, bias: bool):
        if in_channels % groups:
            raise AssertionError(f"in_channels {in_channels} cannot be divisible by groups {groups}")
        if out_channels % groups:
            raise AssertionError(f"out_channels {out_channels} cannot be divisible by groups {groups}")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _two(kernel_size)
        self.groups, self.deformable_groups = groups, deformable_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *self.kernel_size))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        bound = 1.0 / math.sqrt(self.in_channels * self.kernel_size[0] * self.kernel_size[1])
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.zero_()

    def _offset_head(self, planes_per_tap: int) -> nn.Conv2d:
        """The zero-initialised conv that predicts offsets (2 planes per tap) or offsets + mask (3) in the *Pack classes
        (deform_conv.py:243-256, 316-328)."""
        head = nn.Conv2d(self.in_channels, self.deformable_groups * planes_per_tap * self.kernel_size[0] * self.kernel_size[1],
                         kernel_size=self.kernel_size, stride=_two(self.stride), padding=_two(self.padding), bias=True)
        return head

    @staticmethod
    def _zero(head: nn.Conv2d):
        with torch.no_grad():
            head.weight.zero_()
            head.bias.zero_()

    def extra_repr(self):
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, "
                f"padding={self.padding}, dilation={self.dilation}, groups={self.groups}, "
                f"deformable_groups={self.deformable_groups}, bias={self.bias is not None}")


class DeformConv(_DeformableConvModule):
    """deform_conv.py:190-235: ``forward(x, offset)``; stride / padding / dilation stored as pairs; no bias."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=False):
        super().__init__()
        assert not bias
        self.stride, self.padding, self.dilation = _two(stride), _two(padding), _two(dilation)
        self._make_parameters(in_channels, out_channels, kernel_size, groups, deformable_groups, bias=False)

    def forward(self, x, offset):
        return deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups,
                           self.deformable_groups)


class DeformConvPack(DeformConv):
    """deform_conv.py:238-261: the offsets come from ``conv_offset`` applied to the input."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.conv_offset = self._offset_head(2)
        self.init_offset()

    def init_offset(self):
        self._zero(self.conv_offset)

    def forward(self, x):
        return super().forward(x, self.conv_offset(x))


class ModulatedDeformConv(_DeformableConvModule):
    """deform_conv.py:264-308: ``forward(x, offset, mask)``; stride / padding / dilation stored as given (ints)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=True):
        super().__init__()
        self.stride, self.padding, self.dilation, self.with_bias = stride, padding, dilation, bias
        self._make_parameters(in_channels, out_channels, kernel_size, groups, deformable_groups, bias=bias)

    def forward(self, x, offset, mask):
        return modulated_deform_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation,
                                     self.groups, self.deformable_groups)


class ModulatedDeformConvPack(ModulatedDeformConv):
    """deform_conv.py:311-337: ``conv_offset_mask`` predicts 3 planes per tap and group -- two offset thirds and the
    mask logits (sigmoid)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.conv_offset_mask = self._offset_head(3)
        self.init_offset()

    def init_offset(self):
        self._zero(self.conv_offset_mask)

    def forward(self, x):
        planes = self.conv_offset_mask(x)
        n = planes.shape[1] // 3
        return super().forward(x, planes[:, :2 * n], torch.sigmoid(planes[:, 2 * n:]))
