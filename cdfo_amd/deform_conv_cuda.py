"""Stand-in for the reference's pybind11 module ``deform_conv_cuda`` (ops/dcn/src/deform_conv_cuda.cpp:681-695):
the same five entry points with the same positional signatures, backed by libcdfo_hip.so's ``cdfo_dcn_forward``.

The out-parameter convention is kept (caller pre-allocates ``output``; ``columns``/``ones`` scratch tensors are
accepted and ignored -- the HIP kernels fuse sampling and contraction).  The three backward entry points share
``cdfo_dcn_backward`` and keep the reference's conventions: gradient tensors arrive zero-filled,
grad_input / grad_weight / grad_bias are accumulated into, grad_offset / grad_mask are assigned."""
from __future__ import annotations

import ctypes as C

import torch

import os

from . import _lib
from .kernels import _stream, on_device

# Forward arithmetic.  False (default): shapes the fused fast kernel covers (groups == 1, (C/dg) % 4 == 0, Co % 32 == 0,
# Co <= 128) run on split-fp16 MFMA (fp32-grade: <= 2e-5 * max|out| against the C oracle); the kernel scales its operands by
# powers of two taken from max|input| / max|weight| and, should a sampled value still leave the fp16 range (or be
# non-finite), the library re-runs the exact-fp32 kernel over the whole result -- so the fast path is range-safe.
# True (or CDFO_DCN_EXACT=1 in the environment): always the exact-fp32 MFMA kernel (v_mfma_f32_32x32x2_f32, bitwise an
# fp32 fma chain), ~3.5x slower at the alignment module's shape.
EXACT_FP32 = os.environ.get("CDFO_DCN_EXACT", "0") not in ("", "0")
_DEBUG_KEEP_WS = os.environ.get("CDFO_DCN_DEBUG", "0") not in ("", "0")      # developer switch: keep the last workspace alive
_debug_last: list = []


_DTYPES = {torch.float32: 0, torch.float16: 1, torch.float64: 2}     # CDFO_DTYPE_* of include/cdfo_hip.h


def _check_cuda(*ts):
    """Device tensors of ONE of the reference's three dtypes (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
    deform_conv_cuda_kernel.cu:258): returns the library's dtype tag."""
    dt = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise NotImplementedError("deform_conv_cuda (HIP): CPU tensors are not supported")
        if t.dtype not in _DTYPES:
            raise RuntimeError(f'"deform_conv" not implemented for \'{t.dtype}\' (float, double and half are)')
        if dt is not None and t.dtype != dt:
            raise RuntimeError(f"expected scalar type {dt} but found {t.dtype}")
        dt = t.dtype
    return _DTYPES[dt]


def _fwd(input, weight, bias, offset, mask, output, kh, kw, sh, sw, ph, pw, dh, dw, group, dg):
    if not input.is_contiguous():
        raise RuntimeError("input tensor has to be contiguous")      # cpp:493
    if not weight.is_contiguous():
        raise RuntimeError("weight tensor has to be contiguous")     # cpp:494
    dt = _check_cuda(input, weight, bias, offset, mask, output)
    B, Cc, H, W = input.shape
    Co, Ck, kh_, kw_ = weight.shape
    if (kh_, kw_) != (kh, kw):
        raise RuntimeError(f"Input shape and kernel shape wont match: ({kh} x {kw} vs {kh_} x {kw_}).")
    if Cc != Ck * group:
        raise RuntimeError(f"Input shape and kernel channels wont match: ({Cc} vs {Ck * group}).")
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    if tuple(offset.shape) != (B, 2 * dg * kh * kw, Ho, Wo):
        raise RuntimeError(f"invalid offset shape {tuple(offset.shape)}, expected {(B, 2 * dg * kh * kw, Ho, Wo)}")
    if mask is not None and tuple(mask.shape) != (B, dg * kh * kw, Ho, Wo):
        raise RuntimeError(f"invalid mask shape {tuple(mask.shape)}")
    if output.numel() != B * Co * Ho * Wo or not output.is_contiguous():
        raise RuntimeError("output must be a contiguous tensor of B*Co*Ho*Wo elements")
    offset = offset.contiguous()
    mask = None if mask is None else mask.contiguous()
    bias = None if bias is None else bias.contiguous()
    p = lambda t: C.c_void_p(None if t is None else t.data_ptr())  # noqa: E731
    L = _lib.lib()
    with on_device(input):
        # device scratch (plumbing, no arithmetic): what the library asks for these shapes -- the fast kernel's packed
        # operands / the group-planar copy of `input` (fp32), plus the widened operands (half).  EXACT_FP32: only the
        # group-planar copy is offered, which is less than the fast kernel wants, so the exact-fp32 kernel runs.
        nbytes = int(L.cdfo_dcn_workspace_bytes_dt(dt, 0, B, Cc, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, group, dg))
        if nbytes < 0:
            raise RuntimeError("deform_conv_cuda (HIP): unsupported shape")
        if EXACT_FP32 and dt == 0:
            nbytes = input.numel() * 4
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=input.device)
        _lib.check(L.cdfo_dcn_forward_dt(dt, p(input), p(offset), p(mask), p(weight), p(bias), p(output), B, Cc, H, W,
                                         Co, kh, kw, sh, sw, ph, pw, dh, dw, group, dg, p(ws), C.c_longlong(nbytes),
                                         _stream()), "cdfo_dcn_forward_dt")
        if _DEBUG_KEEP_WS:
            _debug_last.clear()
            _debug_last.extend([ws, nbytes])


def deform_conv_forward_cuda(input, weight, offset, output, columns, ones, kW, kH, dW, dH, padW, padH, dilationW,
                             dilationH, group, deformable_group, im2col_step):
    """ops/dcn/src/deform_conv_cuda.cpp:151-156.  Returns 1 like the reference (cpp:257)."""
    _fwd(input, weight, None, offset, None, output, kH, kW, dH, dW, padH, padW, dilationH, dilationW, group,
         deformable_group)
    return 1


def modulated_deform_conv_cuda_forward(input, weight, bias, ones, offset, mask, output, columns, kernel_h, kernel_w,
                                       stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group,
                                       deformable_group, with_bias):
    """ops/dcn/src/deform_conv_cuda.cpp:486-492."""
    _fwd(input, weight, bias if with_bias else None, offset, mask, output, kernel_h, kernel_w, stride_h, stride_w,
         pad_h, pad_w, dilation_h, dilation_w, group, deformable_group)


def _bwd(input, offset, mask, weight, grad_output, grad_input, grad_offset, grad_mask, grad_weight, grad_bias, kh, kw,
         sh, sw, ph, pw, dh, dw, group, dg, scale):
    if not input.is_contiguous():
        raise RuntimeError("input tensor has to be contiguous")      # cpp:574
    if not weight.is_contiguous():
        raise RuntimeError("weight tensor has to be contiguous")     # cpp:575
    dt = _check_cuda(input, offset, mask, weight, grad_output, grad_input, grad_offset, grad_mask, grad_weight, grad_bias)
    B, Cc, H, W = input.shape
    Co, Ck, kh_, kw_ = weight.shape
    if (kh_, kw_) != (kh, kw):
        raise RuntimeError(f"Input shape and kernel shape wont match: ({kh} x {kw} vs {kh_} x {kw_}).")
    if Cc != Ck * group:
        raise RuntimeError(f"Input shape and kernel channels wont match: ({Cc} vs {Ck * group}).")
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    if tuple(offset.shape) != (B, 2 * dg * kh * kw, Ho, Wo):
        raise RuntimeError(f"invalid offset shape {tuple(offset.shape)}, expected {(B, 2 * dg * kh * kw, Ho, Wo)}")
    if mask is not None and tuple(mask.shape) != (B, dg * kh * kw, Ho, Wo):
        raise RuntimeError(f"invalid mask shape {tuple(mask.shape)}")
    if tuple(grad_output.shape) != (B, Co, Ho, Wo):
        raise RuntimeError(f"invalid gradOutput shape {tuple(grad_output.shape)}, expected {(B, Co, Ho, Wo)}")
    for name, t, like in (("grad_input", grad_input, input), ("grad_offset", grad_offset, offset),
                          ("grad_mask", grad_mask, mask), ("grad_weight", grad_weight, weight)):
        if t is not None and (t.numel() != like.numel() or not t.is_contiguous()):
            raise RuntimeError(f"{name} must be a contiguous tensor shaped like its forward counterpart")
    if grad_bias is not None and (grad_bias.numel() != Co or not grad_bias.is_contiguous()):
        raise RuntimeError("grad_bias must be a contiguous tensor of Co elements")
    offset, grad_output = offset.contiguous(), grad_output.contiguous()
    mask = None if mask is None else mask.contiguous()
    p = lambda t: C.c_void_p(None if t is None else t.data_ptr())  # noqa: E731
    L = _lib.lib()
    with on_device(input):
        nbytes = int(L.cdfo_dcn_workspace_bytes_dt(dt, 1, B, Cc, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, group, dg))
        if nbytes < 0:
            raise RuntimeError("deform_conv_cuda (HIP): unsupported shape")
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=input.device)
        _lib.check(L.cdfo_dcn_backward_dt(dt, p(input), p(offset), p(mask), p(weight), p(grad_output), p(grad_input),
                                          p(grad_offset), p(grad_mask), p(grad_weight), p(grad_bias), B, Cc, H, W, Co,
                                          kh, kw, sh, sw, ph, pw, dh, dw, group, dg, float(scale), p(ws),
                                          C.c_longlong(nbytes), _stream()), "cdfo_dcn_backward_dt")


def deform_conv_backward_input_cuda(input, offset, gradOutput, gradInput, gradOffset, weight, columns, kW, kH, dW, dH,
                                    padW, padH, dilationW, dilationH, group, deformable_group, im2col_step):
    """ops/dcn/src/deform_conv_cuda.cpp:260-266.  Returns 1 like the reference (cpp:370)."""
    _bwd(input, offset, None, weight, gradOutput, gradInput, gradOffset, None, None, None, kH, kW, dH, dW, padH, padW,
         dilationH, dilationW, group, deformable_group, 1.0)
    return 1


def deform_conv_backward_parameters_cuda(input, offset, gradOutput, gradWeight, columns, ones, kW, kH, dW, dH, padW,
                                         padH, dilationW, dilationH, group, deformable_group, scale, im2col_step):
    """ops/dcn/src/deform_conv_cuda.cpp:373-378.  ``gradWeight += scale * dW``; returns 1 (cpp:483)."""
    if gradWeight.dim() != 4 or gradWeight.size(1) * group != input.size(1):
        raise RuntimeError("gradWeight must be [Co, C/groups, kH, kW]")
    # the forward weight is not an argument of this entry point (the product only needs its shape)
    _bwd(input, offset, None, gradWeight, gradOutput, None, None, None, gradWeight, None, kH, kW, dH, dW, padH, padW,
         dilationH, dilationW, group, deformable_group, scale)
    return 1


def modulated_deform_conv_cuda_backward(input, weight, bias, ones, offset, mask, columns, grad_input, grad_weight,
                                        grad_bias, grad_offset, grad_mask, grad_output, kernel_h, kernel_w, stride_h,
                                        stride_w, pad_h, pad_w, dilation_h, dilation_w, group, deformable_group,
                                        with_bias):
    """ops/dcn/src/deform_conv_cuda.cpp:566-573."""
    _bwd(input, offset, mask, weight, grad_output, grad_input, grad_offset, grad_mask, grad_weight,
         grad_bias if with_bias else None, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
         group, deformable_group, 1.0)
