"""Thin host wrappers: torch tensors (device memory + stream plumbing only) -> libcdfo_hip.so C-ABI calls.

Activations are fp32 pixel-major tensors of logical shape [B, H, W, C] whose last dim is contiguous; the pixel
pitch ``ld = t.stride(2)`` may exceed C, so a channel slice ``t[..., a:b]`` of a wider buffer is a valid operand
(this is how ``torch.cat`` disappears from the path)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import ConvArgs, check

ACT_NONE, ACT_LRELU, ACT_RELU, ACT_SIGMOID = 0, 1, 2, 3


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk_act(t: torch.Tensor, name: str = "tensor"):
    if not t.is_cuda:
        raise NotImplementedError(f"{name}: the HIP path needs device tensors (no CPU fallback)")
    if t.dtype != torch.float32 or t.dim() != 4 or t.stride(3) != 1:
        raise ValueError(f"{name}: expected fp32 [B,H,W,C] with contiguous channels, got {t.dtype} {tuple(t.shape)} "
                         f"strides {t.stride()}")
    B, H, W, Cc = t.shape
    ld = t.stride(2)
    if (W > 1 and t.stride(1) != W * ld) or (B > 1 and t.stride(0) != H * W * ld):
        raise ValueError(f"{name}: rows/images must be densely packed at pitch ld={ld}, strides {t.stride()}")
    return B, H, W, Cc, ld


def empty_act(B: int, H: int, W: int, Cc: int, device) -> torch.Tensor:
    return torch.empty((B, H, W, Cc), dtype=torch.float32, device=device)


# ----------------------------------------------------------------------------------------------- layout
def nchw_to_nhwc(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    B, Cc, H, W = x.shape
    x = x.contiguous()
    if out is None:
        out = empty_act(B, H, W, Cc, x.device)
    _, _, _, _, ldo = _chk_act(out, "out")
    check(_lib.lib().cdfo_nchw_to_nhwc(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), B, Cc, H, W, ldo,
                                       _stream()), "cdfo_nchw_to_nhwc")
    return out


def nhwc_to_nchw(x: torch.Tensor) -> torch.Tensor:
    B, H, W, Cc, ld = _chk_act(x, "x")
    out = torch.empty((B, Cc, H, W), dtype=torch.float32, device=x.device)
    check(_lib.lib().cdfo_nhwc_to_nchw(C.c_void_p(x.data_ptr()), ld, C.c_void_p(out.data_ptr()), B, Cc, H, W,
                                       _stream()), "cdfo_nhwc_to_nchw")
    return out


# ----------------------------------------------------------------------------------------------- conv
@dataclass
class PackedConv:
    w: torch.Tensor            # packed [ks*ks][Cin/4][CoutP][4]  (optionally [B] of those)
    bias: Optional[torch.Tensor]
    Cout: int
    Cin: int
    ks: int
    CoutP: int
    shuffle2: bool = False
    w_bstride: int = 0


def pack_conv(weight: torch.Tensor, bias: Optional[torch.Tensor], shuffle2: bool = False,
              transposed: bool = False) -> PackedConv:
    """weight: OIHW (or IOHW for a ConvTranspose2d) fp32 on the device."""
    w = weight.detach().contiguous().float()
    if transposed:
        Cin, Cout, ks, _ = w.shape
    else:
        Cout, Cin, ks, _ = w.shape
    CoutP = (Cout + 31) // 32 * 32
    packed = torch.empty(ks * ks * Cin * CoutP, dtype=torch.float32, device=w.device)
    check(_lib.lib().cdfo_pack_conv_weight(C.c_void_p(w.data_ptr()), C.c_void_p(packed.data_ptr()), Cout, Cin, ks,
                                           int(shuffle2), int(transposed), _stream()), "cdfo_pack_conv_weight")
    b = None
    if bias is not None:
        b = bias.detach().contiguous().float()
        if shuffle2:
            cq = Cout // 4
            b = b.view(cq, 4).t().contiguous().view(-1)
    return PackedConv(packed, b, Cout, Cin, ks, CoutP, shuffle2)


def conv(srcs: Sequence[torch.Tensor], pc: PackedConv, *, stride: int = 1, pad: int = 0, act: int = ACT_NONE,
         res1: Optional[torch.Tensor] = None, res2: Optional[torch.Tensor] = None,
         out: Optional[torch.Tensor] = None) -> torch.Tensor:
    if isinstance(srcs, torch.Tensor):
        srcs = [srcs]
    a = ConvArgs()
    B = H = W = None
    cin = 0
    for i, s in enumerate(srcs):
        b_, h_, w_, c_, ld_ = _chk_act(s, f"src{i}")
        if B is None:
            B, H, W = b_, h_, w_
        elif (B, H, W) != (b_, h_, w_):
            raise ValueError("conv sources disagree on B/H/W")
        a.src[i] = s.data_ptr()
        a.ld[i] = ld_
        a.cs[i] = c_
        cin += c_
    if cin != pc.Cin:
        raise ValueError(f"conv: sources have {cin} channels, weight expects {pc.Cin}")
    a.nsrc = len(srcs)
    Ho = (H + 2 * pad - pc.ks) // stride + 1
    Wo = (W + 2 * pad - pc.ks) // stride + 1
    a.B, a.H, a.W, a.Ho, a.Wo = B, H, W, Ho, Wo
    a.ks, a.stride, a.pad = pc.ks, stride, pad
    a.Cin, a.Cout, a.CoutP = pc.Cin, pc.Cout, pc.CoutP
    a.w = pc.w.data_ptr()
    a.w_bstride = pc.w_bstride
    a.bias = _p(pc.bias)
    a.act = act
    if pc.shuffle2:
        if out is None:
            out = empty_act(B, 2 * Ho, 2 * Wo, pc.Cout // 4, srcs[0].device)
        a.store_mode = 1
    else:
        if out is None:
            out = empty_act(B, Ho, Wo, pc.Cout, srcs[0].device)
        a.store_mode = 0
    _, _, _, _, a.ldo = _chk_act(out, "out")
    a.out = out.data_ptr()
    for nm, r in (("res1", res1), ("res2", res2)):
        if r is not None:
            rb, rh, rw, rc, rld = _chk_act(r, nm)
            if (rb, rh, rw) != (B, Ho, Wo) or rc < pc.Cout:
                raise ValueError(f"{nm} shape {tuple(r.shape)} does not match the conv output")
            setattr(a, nm, r.data_ptr())
            setattr(a, "ldr" + nm[-1], rld)
    a.prec = 0
    check(_lib.lib().cdfo_conv_igemm(C.byref(a), _stream()), "cdfo_conv_igemm")
    return out
