"""Thin host wrappers: torch tensors (device memory + stream plumbing only) -> libcdfo_hip.so C-ABI calls.

Activations are fp32 pixel-major tensors of logical shape [B, H, W, C] whose last dim is contiguous; the pixel
pitch ``ld = t.stride(2)`` may exceed C, so a channel slice ``t[..., a:b]`` of a wider buffer is a valid operand
(this is how ``torch.cat`` disappears from the path)."""
from __future__ import annotations

import contextlib
import os
import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import CdfoError, ConvArgs, check

ACT_NONE, ACT_LRELU, ACT_RELU, ACT_SIGMOID = 0, 1, 2, 3
PREC_F32, PREC_BF16X3, PREC_BF16, PREC_FP16X2, PREC_FP16, PREC_FP16X1 = 0, 1, 2, 3, 4, 5


def _stream() -> C.c_void_p:
    """The CURRENT device's current stream.  Every public entry point (module forwards, the DCN operator, the metrics)
    makes its operands' device current first (`on_device`), and `_chk_act` refuses operands of another device, so a
    launch never pairs one device's stream with another device's pointers."""
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def on_device(t: torch.Tensor):
    """Context manager: make `t`'s device the current one (stream lookup, per-device caches of libcdfo_hip.so)."""
    if not t.is_cuda:
        raise NotImplementedError("the HIP path needs device tensors (there is no CPU fallback)")
    return torch.cuda.device(t.device)


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk_act(t: torch.Tensor, name: str = "tensor", dtype=torch.float32):
    if not t.is_cuda:
        raise NotImplementedError(f"{name}: the HIP path needs device tensors (no CPU fallback)")
    if t.device.index != torch.cuda.current_device():
        raise CdfoError(f"{name} lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}: "
                        "enter kernels.on_device(tensor) (the module forwards do) before calling the wrappers")
    if t.dtype != dtype or t.dim() != 4 or t.stride(3) != 1:
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


def range_probe(x: torch.Tensor, slots: torch.Tensor) -> None:
    """slots (int32[2], zeroed by the caller): [0] = bits of max finite |x|, [1] = 1 if x holds a NaN / infinity."""
    if not x.is_contiguous() or x.dtype != torch.float32 or x.numel() % 4:
        raise ValueError("range_probe: dense fp32 tensor with a multiple of 4 elements expected")
    check(_lib.lib().cdfo_range_probe(C.c_void_p(x.data_ptr()), C.c_longlong(x.numel()), C.c_void_p(slots.data_ptr()),
                                      _stream()), "cdfo_range_probe")


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
    wq: Optional[torch.Tensor] = None   # split-bf16 packing for cdfo_conv3x3_bf16 (3x3, Cout % 64 == 0)
    tap_mask: Optional[torch.Tensor] = None   # int32 [Cin/16]: bit t set = tap t of that chunk has weights
    wh: Optional[torch.Tensor] = None   # fp16 packing (single block) for PREC_FP16X2
    CoutP16: int = 0                    # padded output channels of the 16-bit packings (multiple of 64)
    wh_sparse: Optional[torch.Tensor] = None   # fp16 packing of the four active taps per chunk (conv_ring + tap_mask)
    ww: Optional[torch.Tensor] = None   # Winograd F(2,3) fp16 image for conv3x3_wino (3x3, Cin = 64, Cout % 128 == 0)


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
    pc = PackedConv(packed, b, Cout, Cin, ks, CoutP, shuffle2)
    if ks == 3 and not transposed and not shuffle2 and Cout % 4 == 0 and Cin % 16 == 0:
        cp16 = (Cout + 63) // 64 * 64          # thin outputs (UDSA 64->16) ride on a 64-wide tile, masked in the epilogue
        pc.CoutP16 = cp16
        wq = torch.empty(2 * (Cin // 16) * 18 * cp16 * 8, dtype=torch.bfloat16, device=w.device)
        check(_lib.lib().cdfo_pack_conv3x3_bf16(C.c_void_p(w.data_ptr()), C.c_void_p(wq.data_ptr()), Cout, Cin,
                                                _stream()), "cdfo_pack_conv3x3_bf16")
        pc.wq = wq
        wh = torch.empty((Cin // 16) * 18 * cp16 * 8, dtype=torch.float16, device=w.device)
        check(_lib.lib().cdfo_pack_conv3x3_f16(C.c_void_p(w.data_ptr()), C.c_void_p(wh.data_ptr()), Cout, Cin, _stream()),
              "cdfo_pack_conv3x3_f16")
        pc.wh = wh
        if Cin == 64 and Cout % 128 == 0:
            ww = torch.empty(Cout * 64 * 12, dtype=torch.float16, device=w.device)
            check(_lib.lib().cdfo_pack_conv3x3_wino(C.c_void_p(w.data_ptr()), C.c_void_p(ww.data_ptr()), Cout, _stream()),
                  "cdfo_pack_conv3x3_wino")
            pc.ww = ww
    return pc


def conv(srcs: Sequence[torch.Tensor], pc: PackedConv, *, stride: int = 1, pad: int = 0, act: int = ACT_NONE,
         res1: Optional[torch.Tensor] = None, res2: Optional[torch.Tensor] = None,
         out: Optional[torch.Tensor] = None, prec: int = PREC_F32,
         ln: Optional[tuple] = None, s2d: bool = False, out_f16: bool = False, ln_out: Optional[tuple] = None,
         res2_scale: Optional[torch.Tensor] = None, cp16_out: bool = False):
    """cp16_out (1x1, Cout = 64, 16-bit modes): also the fp16 chunk-planar copy [B,4,H,W,16] of the result (to_cp16 of the returned
    tensor) from the same kernel; the call then returns the pair (out, copy).
    res2_scale [B,H,W]: res2 enters the sum as res2 * res2_scale[pixel] (streaming 1x1 kernel only; raises elsewhere).
    ln_out = (gamma, beta) (1x1, Cout = 64, 16-bit modes): also LayerNorm64 of the RESULT as fp16 hi | lo planes
    [B,8,H,W,16] (layernorm64_hl of the returned tensor); the call then returns the pair (out, planes)."""
    if isinstance(srcs, torch.Tensor):
        srcs = [srcs]
    a = ConvArgs()
    B = H = W = None
    cin = 0
    src_f16 = srcs[0].dtype == torch.float16       # Block_'s fp16 body intermediate (single source, 1-pass fp16 MFMA)
    if src_f16:
        prec = PREC_FP16
    elif prec == PREC_FP16:
        raise ValueError("PREC_FP16 needs an fp16 source tensor")
    odt = torch.float16 if out_f16 else torch.float32
    for i, s in enumerate(srcs):
        b_, h_, w_, c_, ld_ = _chk_act(s, f"src{i}", torch.float16 if src_f16 else torch.float32)
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
    elif s2d:
        if out is None:
            out = torch.empty((B, Ho // 2, Wo // 2, 4 * pc.Cout), dtype=odt, device=srcs[0].device)
        a.store_mode = 2
    else:
        if out is None:
            out = torch.empty((B, Ho, Wo, pc.Cout), dtype=odt, device=srcs[0].device)
        a.store_mode = 0
    _, _, _, _, a.ldo = _chk_act(out, "out", odt)
    a.src_f16, a.out_f16 = int(src_f16), int(out_f16)
    a.out = out.data_ptr()
    for nm, r in (("res1", res1), ("res2", res2)):
        if r is not None:
            rb, rh, rw, rc, rld = _chk_act(r, nm)
            if (rb, rh, rw) != (B, Ho, Wo) or rc < pc.Cout:
                raise ValueError(f"{nm} shape {tuple(r.shape)} does not match the conv output")
            setattr(a, nm, r.data_ptr())
            setattr(a, "ldr" + nm[-1], rld)
    if res2_scale is not None:
        if res2 is None or tuple(res2_scale.shape) != (B, Ho, Wo) or not res2_scale.is_contiguous() or res2_scale.dtype != torch.float32:
            raise ValueError("conv: res2_scale must be a contiguous fp32 [B,H,W] plane next to res2")
        a.res2_pixscale = res2_scale.data_ptr()
    if (prec != PREC_F32 and pc.wq is not None and stride == 1 and pad == 1 and pc.w_bstride == 0):
        a.prec = prec
        a.CoutP = pc.CoutP16
        a.w = (pc.wh if prec in (PREC_FP16X2, PREC_FP16, PREC_FP16X1) else pc.wq).data_ptr()
        a.tap_mask = _p(pc.tap_mask)
        check(_lib.lib().cdfo_conv3x3_bf16(C.byref(a), _stream()), "cdfo_conv3x3_bf16")
        return out
    if src_f16 or out_f16:
        raise ValueError("fp16 tensors are only supported by the 16-bit MFMA 3x3 kernel")
    a.prec = 0
    if ln is not None:
        a.ln_gamma, a.ln_beta = ln[0].data_ptr(), ln[1].data_ptr()
    if (prec != PREC_F32 and pc.ks == 1 and stride == 1 and pad == 0 and pc.CoutP % 64 == 0 and pc.CoutP <= 256
            and all(s.shape[3] % 64 == 0 for s in srcs)):
        if cp16_out:
            if ln_out is not None or ln is not None or pc.Cout != 64 or pc.CoutP != 64:
                raise ValueError("conv: cp16_out needs a plain 64-channel 1x1 convolution")
            copy = torch.empty((B, 4, Ho, Wo, 16), dtype=torch.float16, device=out.device)
            a.out2_cp16 = copy.data_ptr()
            check(_lib.lib().cdfo_conv1x1_bf16x3(C.byref(a), _stream()), "cdfo_conv1x1_bf16x3")
            return out, copy
        planes = None
        if ln_out is not None and pc.Cout == 64 and pc.CoutP == 64 and ln is None:
            planes = torch.empty((B, 8, Ho, Wo, 16), dtype=torch.float16, device=out.device)
            a.out2_cp16, a.ln_gamma, a.ln_beta = planes.data_ptr(), ln_out[0].data_ptr(), ln_out[1].data_ptr()
        check(_lib.lib().cdfo_conv1x1_bf16x3(C.byref(a), _stream()), "cdfo_conv1x1_bf16x3")
        if ln_out is not None:
            return out, (planes if planes is not None else layernorm64_hl(out, ln_out[0], ln_out[1]))
        return out
    check(_lib.lib().cdfo_conv_igemm(C.byref(a), _stream()), "cdfo_conv_igemm")
    if ln_out is not None:
        return out, layernorm64_hl(out, ln_out[0], ln_out[1])
    if cp16_out:
        return out, to_cp16(out)
    return out


def _vp(t):
    return C.c_void_p(None if t is None else t.data_ptr())


@contextlib.contextmanager
def cu_limit(n: int):
    """Persistent kernels launched by this thread inside the block fill at most n compute units (cdfo_set_cu_limit)."""
    prev = _lib.lib().cdfo_set_cu_limit(int(n))
    try:
        yield
    finally:
        _lib.lib().cdfo_set_cu_limit(prev)


def conv_offset_mask(src: torch.Tensor, pc: PackedConv, offset: torch.Tensor, mask: torch.Tensor, flow: torch.Tensor, mag: float,
                     accumulate: bool, prec: int) -> None:
    """MVDualAttAlignment's conv_offset[2] (3x3, 64 -> 27 dg) with the module's offset / mask assembly (arch.py:3336-3350) as its
    epilogue (CDFO_STORE_OFFMASK): src pixel-major [B,H,W,64]; offset [B,18 dg,H,W] / mask [B,9 dg,H,W] NCHW = the DCN operator's
    inputs; flow [B,2,H,W] contiguous.  accumulate=False: the first head (offset = mag tanh + flipped flow, mask = raw sums);
    accumulate=True: the second head in place (offset += mag tanh, mask = sigmoid(mask + sums)).  W % 4 == 0, 16-bit modes only."""
    B, H, W, Cc, ld = _chk_act(src)
    third = pc.Cout // 3
    if (pc.ks != 3 or Cc != pc.Cin or pc.Cout % 3 or pc.wq is None or W % 4 or prec not in (PREC_BF16X3, PREC_FP16X2, PREC_FP16X1)
            or tuple(offset.shape) != (B, 2 * third, H, W) or tuple(mask.shape) != (B, third, H, W) or tuple(flow.shape) != (B, 2, H, W)
            or not (offset.is_contiguous() and mask.is_contiguous() and flow.is_contiguous())
            or any(t.dtype != torch.float32 for t in (offset, mask, flow))):
        raise ValueError("conv_offset_mask: unsupported configuration")
    a = ConvArgs()
    a.src[0], a.ld[0], a.cs[0], a.nsrc = src.data_ptr(), ld, Cc, 1
    a.B, a.H, a.W, a.Ho, a.Wo = B, H, W, H, W
    a.ks, a.stride, a.pad = 3, 1, 1
    a.Cin, a.Cout, a.CoutP = pc.Cin, pc.Cout, pc.CoutP16
    a.w = (pc.wh if prec in (PREC_FP16X2, PREC_FP16X1) else pc.wq).data_ptr()
    a.bias = _p(pc.bias)
    a.act, a.prec, a.store_mode = ACT_NONE, prec, 4
    a.out, a.ldo = offset.data_ptr(), 4
    a.mask_out, a.flow, a.flow_bstride = mask.data_ptr(), flow.data_ptr(), 2 * H * W
    a.off_mag, a.off_accumulate = float(mag), int(accumulate)
    check(_lib.lib().cdfo_conv3x3_bf16(C.byref(a), _stream()), "cdfo_conv3x3_bf16 (offset / mask epilogue)")


def pack_conv_n16(weight: torch.Tensor) -> torch.Tensor:
    """[16, 64, 3, 3] fp32 -> the bf16 hi | lo image of cdfo_conv3x3_c64_n16: [18 K steps = tap*2 + half][hi|lo][lane = kg*16 + m][8]
    with element j = W[m][32 half + 8 kg + j][tap] (parameter-sized torch arithmetic, once per weight version)."""
    if tuple(weight.shape) != (16, 64, 3, 3):
        raise ValueError("pack_conv_n16: a [16, 64, 3, 3] weight expected")
    w = weight.detach().float().permute(2, 3, 1, 0).reshape(9, 2, 4, 8, 16).permute(0, 1, 2, 4, 3).reshape(18, 64, 8)
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    return torch.stack([hi, lo], 1).contiguous()              # [18, 2, 64, 8] bf16 = 36,864 bytes


def conv3x3_n16(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], act: int = ACT_NONE) -> torch.Tensor:
    """3x3 / stride 1 / pad 1 convolution 64 -> 16 (split-bf16, fp32-grade): x [B,H,W,64] (channel slices allowed) -> [B,H,W,16]."""
    B, H, W, Cc, ld = _chk_act(x)
    if Cc != 64 or w_packed.dtype != torch.bfloat16 or w_packed.numel() != 18 * 2 * 64 * 8:
        raise ValueError("conv3x3_n16: 64 input channels and a pack_conv_n16 weight image expected")
    out = empty_act(B, H, W, 16, x.device)
    check(_lib.lib().cdfo_conv3x3_c64_n16(_vp(x), ld, B, H, W, _vp(w_packed), _vp(bias), act, _vp(out), 16, _stream()),
          "cdfo_conv3x3_c64_n16")
    return out


def to_cp16(x: torch.Tensor) -> torch.Tensor:
    """fp32 pixel-major [B,H,W,C] -> fp16 chunk-planar [B, C/16, H, W, 16] (the source layout of conv3x3_ws)."""
    B, H, W, Cc, ld = _chk_act(x)
    out = torch.empty((B, Cc // 16, H, W, 16), dtype=torch.float16, device=x.device)
    check(_lib.lib().cdfo_to_cp16(_vp(x), ld, B, C.c_longlong(H * W), Cc, _vp(out), _stream()), "cdfo_to_cp16")
    return out


def conv3x3_ws(src: torch.Tensor, pc: PackedConv, *, act: int = ACT_NONE, s2d: bool = False,
               out: Optional[torch.Tensor] = None, dbg: int = 0, clk: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Block_.body[0]-shaped convolution (3x3, 64 input channels, Cout % 64 == 0) on the weights-stationary kernel.
    src: fp16 chunk-planar [B,4,H,W,16]; result: fp16 chunk-planar [B,Cout/16,H,W,16], or with s2d its space-to-depth
    form [B,4*Cout/16,H/2,W/2,16] (chunk = phase*Cout/16 + channel/16, phase = (y&1)*2 + (x&1))."""
    if not src.is_cuda:
        raise NotImplementedError("conv3x3_ws: the HIP path needs device tensors (no CPU fallback)")
    if src.dtype != torch.float16 or src.dim() != 5 or src.shape[1] != 4 or src.shape[4] != 16 or not src.is_contiguous():
        raise ValueError(f"conv3x3_ws: expected a contiguous fp16 [B,4,H,W,16] source, got {src.dtype} {tuple(src.shape)}")
    if pc.wh is None or pc.Cin != 64 or pc.ks != 3 or pc.Cout % 64:
        raise ValueError("conv3x3_ws: needs a 3x3 weight with 64 input channels and Cout % 64 == 0")
    B, _, H, W, _ = src.shape
    shape = (B, pc.Cout // 4, H // 2, W // 2, 16) if s2d else (B, pc.Cout // 16, H, W, 16)
    if out is None:
        out = torch.empty(shape, dtype=torch.float16, device=src.device)
    elif out.shape != shape or out.dtype != torch.float16 or not out.is_contiguous():
        raise ValueError(f"conv3x3_ws: out must be a contiguous fp16 tensor of shape {shape}")
    # the kernel addresses its source with 32-bit buffer offsets (< 2 GiB per launch): split the batch if needed
    per_img = 4 * H * W * 32
    step = max(1, min(B, ((1 << 31) - 1) // per_img))
    if per_img >= (1 << 31):
        raise ValueError(f"conv3x3_ws: one {H}x{W} image exceeds the 2 GiB source limit of a launch")
    for b0 in range(0, B, step):
        nb = min(step, B - b0)
        check(_lib.lib().cdfo_conv3x3_c64_ws(_vp(src[b0:b0 + nb]), nb, H, W, _vp(pc.wh), pc.CoutP16, _vp(pc.bias), pc.Cout,
                                             act, _vp(out[b0:b0 + nb]), 2 if s2d else 0, dbg, _vp(clk), _stream()),
              "cdfo_conv3x3_c64_ws")
    return out


def wino_enabled() -> bool:
    """CDFO_WINO=0 keeps Block_.body[0] on the direct weights-stationary kernel (developer A/B switch)."""
    return os.environ.get("CDFO_WINO", "1") != "0"


def halfsplit_to_rows(t: torch.Tensor) -> torch.Tensor:
    """[B, planes, H, W, 16] stored with half-split rows ([H][2][W][8], CDFO_STORE_S2D_HS) -> the natural [B, planes, H, W, 16] (tests)."""
    B, P, H, W, _ = t.shape
    return t.reshape(B, P, H, 2, W, 8).permute(0, 1, 2, 4, 3, 5).reshape(B, P, H, W, 16)


def conv3x3_wino(src: torch.Tensor, pc: PackedConv, *, act: int = ACT_NONE, s2d: bool = False,
                 out: Optional[torch.Tensor] = None, dbg: int = 0, halfsplit: bool = False, clk: Optional[torch.Tensor] = None) -> torch.Tensor:
    """conv3x3_ws's operands and result on the row-streaming Winograd F(2,3) kernel (cdfo_conv3x3_c64_wino): Cout % 128 == 0, W even."""
    if not src.is_cuda:
        raise NotImplementedError("conv3x3_wino: the HIP path needs device tensors (no CPU fallback)")
    if src.dtype != torch.float16 or src.dim() != 5 or src.shape[1] != 4 or src.shape[4] != 16 or not src.is_contiguous():
        raise ValueError(f"conv3x3_wino: expected a contiguous fp16 [B,4,H,W,16] source, got {src.dtype} {tuple(src.shape)}")
    if pc.ww is None:
        raise ValueError("conv3x3_wino: needs a 3x3 weight with 64 input channels and Cout % 128 == 0")
    B, _, H, W, _ = src.shape
    shape = (B, pc.Cout // 4, H // 2, W // 2, 16) if s2d else (B, pc.Cout // 16, H, W, 16)
    if out is None:
        out = torch.empty(shape, dtype=torch.float16, device=src.device)
    elif out.shape != shape or out.dtype != torch.float16 or not out.is_contiguous():
        raise ValueError(f"conv3x3_wino: out must be a contiguous fp16 tensor of shape {shape}")
    if halfsplit and not s2d:
        raise ValueError("conv3x3_wino: half-split rows exist for the space-to-depth store only")
    mode = (5 if halfsplit else 2) if s2d else 0
    if dbg:      # developer ablations (tools/bench_wino.py): wrong results by construction
        check(_lib.lib().cdfo_conv3x3_c64_wino_dbg(_vp(src), B, H, W, _vp(pc.ww), _vp(pc.bias), pc.Cout, act, _vp(out), mode,
                                                   dbg, _vp(clk), _stream()), "cdfo_conv3x3_c64_wino_dbg")
        return out
    check(_lib.lib().cdfo_conv3x3_c64_wino(_vp(src), B, H, W, _vp(pc.ww), _vp(pc.bias), pc.Cout, act, _vp(out), mode,
                                           _stream()), "cdfo_conv3x3_c64_wino")
    return out


def wino_up2_enabled() -> bool:
    """CDFO_WINO_UP2=0 keeps Block_'s x2 branch on a materialised double-resolution source (developer A/B switch)."""
    return wino_enabled() and os.environ.get("CDFO_WINO_UP2", "1") != "0"


def conv3x3_wino_up2(src_lr: torch.Tensor, pc: PackedConv, *, act: int = ACT_NONE, halfsplit: bool = False) -> torch.Tensor:
    """Block_.body[0] on the bilinear x2 of src_lr (fp16 chunk-planar [B,4,h,w,16], h and w even) without materialising it: the
    interpolation is folded into the Winograd kernel's input transform (cdfo_conv3x3_c64_wino_up2).  Result: the space-to-depth form
    [B, 4*Cout/16, h, w, 16] of the [B, Cout/16, 2h, 2w, 16] convolution output, as conv3x3_wino(..., s2d=True)."""
    if not src_lr.is_cuda:
        raise NotImplementedError("conv3x3_wino_up2: the HIP path needs device tensors (no CPU fallback)")
    if src_lr.dtype != torch.float16 or src_lr.dim() != 5 or src_lr.shape[1] != 4 or src_lr.shape[4] != 16 or not src_lr.is_contiguous():
        raise ValueError(f"conv3x3_wino_up2: expected a contiguous fp16 [B,4,h,w,16] source, got {src_lr.dtype} {tuple(src_lr.shape)}")
    B, _, h, w, _ = src_lr.shape
    if pc.ww is None or h % 2 or w % 2:
        raise ValueError("conv3x3_wino_up2: needs a Winograd weight image (Cout % 128 == 0) and even low-resolution sizes")
    out = torch.empty((B, pc.Cout // 4, h, w, 16), dtype=torch.float16, device=src_lr.device)
    check(_lib.lib().cdfo_conv3x3_c64_wino_up2(_vp(src_lr), B, 2 * h, 2 * w, _vp(pc.ww), _vp(pc.bias), pc.Cout, act, _vp(out),
                                               5 if halfsplit else 2, _stream()), "cdfo_conv3x3_c64_wino_up2")
    return out


def conv3x3_body0(src: torch.Tensor, pc: PackedConv, *, act: int = ACT_NONE, s2d: bool = False) -> torch.Tensor:
    """Block_.body[0]-shaped convolution (fp16 chunk-planar in and out): the Winograd F(2,3) kernel where it applies (Cout % 128 == 0,
    even width, one image of source / result below 2 GiB, CDFO_WINO != 0), the direct weights-stationary kernel otherwise."""
    _, _, H, W, _ = src.shape
    if pc.ww is not None and wino_enabled() and W % 2 == 0 and (not s2d or H % 2 == 0) and H * W * 2 * pc.Cout < (1 << 31):
        plain_st = (not s2d) and os.environ.get("CDFO_WINO_NT", "1") == "0"    # developer A/B: the 1x launches WITHOUT non-temporal stores
        return conv3x3_wino(src, pc, act=act, s2d=s2d, dbg=8192 if plain_st else 0)
    return conv3x3_ws(src, pc, act=act, s2d=s2d)


def conv_offset_mask_ws_fits(B: int, H: int, W: int, third: int) -> bool:
    """The 32-bit addressing limits of cdfo_conv3x3_c64_ws_offmask: the whole fp16 source and one image's offset planes below 2 GiB."""
    return 4 * H * W * 32 * B < (1 << 31) and H * W * 2 * third * 4 < (1 << 31)


def conv_offset_mask_ws(src: torch.Tensor, pc: PackedConv, offset: torch.Tensor, mask: torch.Tensor, flow: torch.Tensor, mag: float,
                        accumulate: bool) -> None:
    """conv_offset_mask on the weights-stationary kernel (single-pass fp16 operands): src fp16 chunk-planar [B,4,H,W,16]; the rest as
    conv_offset_mask.  H even, 18 dg a multiple of 32 (dg = 16: 288)."""
    if src.dtype != torch.float16 or src.dim() != 5 or src.shape[1] != 4 or src.shape[4] != 16 or not src.is_contiguous():
        raise ValueError(f"conv_offset_mask_ws: expected a contiguous fp16 [B,4,H,W,16] source, got {src.dtype} {tuple(src.shape)}")
    B, _, H, W, _ = src.shape
    third = pc.Cout // 3
    if (pc.wh is None or pc.Cin != 64 or pc.ks != 3 or pc.Cout % 3 or H % 2 or (2 * third) % 32 or pc.Cout % 8
            or tuple(offset.shape) != (B, 2 * third, H, W) or tuple(mask.shape) != (B, third, H, W) or tuple(flow.shape) != (B, 2, H, W)
            or not (offset.is_contiguous() and mask.is_contiguous() and flow.is_contiguous())
            or any(t.dtype != torch.float32 for t in (offset, mask, flow)) or not conv_offset_mask_ws_fits(B, H, W, third)):
        raise ValueError("conv_offset_mask_ws: unsupported configuration")
    check(_lib.lib().cdfo_conv3x3_c64_ws_offmask(_vp(src), B, H, W, _vp(pc.wh), pc.CoutP16, _vp(pc.bias), pc.Cout, _vp(offset), _vp(mask),
                                                 _vp(flow), C.c_longlong(2 * H * W), float(mag), int(accumulate), _stream()),
          "cdfo_conv3x3_c64_ws_offmask")


def conv3x3_ws_res(src: torch.Tensor, pc: PackedConv, *, res1: torch.Tensor, res2: Optional[torch.Tensor] = None,
                   act: int = ACT_NONE, out: Optional[torch.Tensor] = None,
                   out2_cp16: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Residual form of conv3x3_ws: fp32 pixel-major act(conv + bias) + res1 (+ res2); optionally also the fp16
    chunk-planar copy of the result (out2_cp16 [B,Cout/16,H,W,16])."""
    if not src.is_cuda:
        raise NotImplementedError("conv3x3_ws_res: the HIP path needs device tensors (no CPU fallback)")
    if src.dtype != torch.float16 or src.dim() != 5 or src.shape[1] != 4 or src.shape[4] != 16 or not src.is_contiguous():
        raise ValueError(f"conv3x3_ws_res: expected a contiguous fp16 [B,4,H,W,16] source, got {src.dtype} {tuple(src.shape)}")
    if pc.wh is None or pc.Cin != 64 or pc.ks != 3 or pc.Cout % 64:
        raise ValueError("conv3x3_ws_res: needs a 3x3 weight with 64 input channels and Cout % 64 == 0")
    B, _, H, W, _ = src.shape
    if out is None:
        out = empty_act(B, H, W, pc.Cout, src.device)
    _, _, _, _, ldo = _chk_act(out, "out")
    lds = []
    for nm, t in (("res1", res1), ("res2", res2)):
        if t is None:
            lds.append(0)
            continue
        rb, rh, rw, rc, rld = _chk_act(t, nm)
        if (rb, rh, rw) != (B, H, W) or rc < pc.Cout:
            raise ValueError(f"{nm} shape {tuple(t.shape)} does not match the conv output")
        lds.append(rld)
    if out2_cp16 is not None and (out2_cp16.dtype != torch.float16 or not out2_cp16.is_contiguous()
                                  or tuple(out2_cp16.shape) != (B, pc.Cout // 16, H, W, 16)):
        raise ValueError("conv3x3_ws_res: out2_cp16 must be a contiguous fp16 [B,Cout/16,H,W,16] tensor")
    per_img = 4 * H * W * 32
    if per_img >= (1 << 31):
        raise ValueError(f"conv3x3_ws_res: one {H}x{W} image exceeds the 2 GiB source limit of a launch")
    step = max(1, min(B, ((1 << 31) - 1) // per_img))
    for b0 in range(0, B, step):
        sl = slice(b0, min(B, b0 + step))
        check(_lib.lib().cdfo_conv3x3_c64_ws_res(
            _vp(src[sl]), sl.stop - b0, H, W, _vp(pc.wh), pc.CoutP16, _vp(pc.bias), pc.Cout, act, _vp(out[sl]), ldo,
            _vp(res1[sl]), lds[0], _vp(None if res2 is None else res2[sl]), lds[1],
            _vp(None if out2_cp16 is None else out2_cp16[sl]), _stream()), "cdfo_conv3x3_c64_ws_res")
    return out


def from_cp16(t: torch.Tensor) -> torch.Tensor:
    """chunk-planar [B,C/16,H,W,16] -> pixel-major [B,H,W,C] (torch ops; tests and tools only)."""
    B, nc, H, W, _ = t.shape
    return t.permute(0, 2, 3, 1, 4).reshape(B, H, W, nc * 16)


def sparse_taps_f16(pc: PackedConv) -> torch.Tensor:
    """fp16 packing [Cin/16][9][2][CoutP][8] + tap_mask -> [Cin/16][4][2][CoutP][8]: each chunk's four active taps in
    ascending order (the weight layout cdfo_conv3x3_ring reads when a tap mask is given)."""
    nc = pc.Cin // 16
    masks = pc.tap_mask.tolist()
    idx = []
    for m in masks:
        taps = [t for t in range(9) if (m >> t) & 1]
        if m not in (0x1B, 0x36, 0xD8, 0x1B0):
            raise ValueError("sparse_taps_f16: every chunk's active taps must be a 2x2 window of the 3x3 stencil")
        idx.append(taps)
    idx = torch.tensor(idx, dtype=torch.long, device=pc.wh.device)                  # [nc,4]
    w = pc.wh.view(nc, 9, 2 * pc.CoutP16 * 8)
    return torch.gather(w, 1, idx[:, :, None].expand(nc, 4, w.shape[2])).contiguous().view(-1)


def conv_ring(src: torch.Tensor, pc: PackedConv, *, act: int = ACT_NONE, res1: Optional[torch.Tensor] = None,
              res2: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
              out_f16: bool = False, out2_cp16: Optional[torch.Tensor] = None,
              res_up2: Optional[torch.Tensor] = None, plane_wrap: int = 0, dbg: int = 0,
              out2_hl: bool = False, src_halfsplit: bool = False) -> torch.Tensor:
    """3x3/s1/p1 convolution of an fp16 chunk-planar source [B,Cin/16,H,W,16] on the LDS-DMA ring kernel
    (Block_.body[2] and the composed stride-2 convolution).  Result: pixel-major fp32 (or fp16) [B,H,W,Cout]."""
    if not src.is_cuda:
        raise NotImplementedError("conv_ring: the HIP path needs device tensors (no CPU fallback)")
    if src.dtype != torch.float16 or src.dim() != 5 or src.shape[4] != 16 or not src.is_contiguous():
        raise ValueError(f"conv_ring: expected a contiguous fp16 [B,C/16,H,W,16] source, got {src.dtype} {tuple(src.shape)}")
    B, nc, H, W, _ = src.shape
    if pc.wh is None or pc.ks != 3 or (pc.Cin != nc * 16 and not (plane_wrap == nc and pc.Cin % 16 == 0 and pc.Cin > nc * 16)):
        raise ValueError("conv_ring: weight does not match the source")
    a = ConvArgs()
    a.src[0], a.ld[0], a.cs[0], a.nsrc = src.data_ptr(), 16, pc.Cin, 1
    a.B, a.H, a.W, a.Ho, a.Wo = B, H, W, H, W
    a.ks, a.stride, a.pad = 3, 1, 1
    a.Cin, a.Cout, a.CoutP = pc.Cin, pc.Cout, pc.CoutP16
    if pc.tap_mask is not None:
        if pc.wh_sparse is None:
            pc.wh_sparse = sparse_taps_f16(pc)
        a.w, a.tap_mask = pc.wh_sparse.data_ptr(), pc.tap_mask.data_ptr()
    else:
        a.w = pc.wh.data_ptr()
    a.bias, a.act = _p(pc.bias), act
    odt = torch.float16 if out_f16 else torch.float32
    if out is None:
        out = torch.empty((B, H, W, pc.Cout), dtype=odt, device=src.device)
    _, _, _, _, a.ldo = _chk_act(out, "out", odt)
    a.out, a.store_mode, a.prec = out.data_ptr(), 0, PREC_FP16 | (dbg << 8)
    a.src_f16, a.out_f16 = 1, int(out_f16)
    a.src_plane_wrap = plane_wrap
    a.src_halfsplit = int(src_halfsplit)
    for nm, r in (("res1", res1), ("res2", None if dbg & 16 else res2)):
        if r is not None:
            rb, rh, rw, rc, rld = _chk_act(r, nm)
            if (rb, rh, rw) != (B, H, W) or rc < pc.Cout:
                raise ValueError(f"{nm} shape {tuple(r.shape)} does not match the conv output")
            setattr(a, nm, r.data_ptr())
            setattr(a, "ldr" + nm[-1], rld)
    if res_up2 is not None:         # half-resolution residual, added after bilinear x2
        rb, rh, rw, rc, rld = _chk_act(res_up2, "res_up2")
        if (rb, rh * 2, rw * 2) != (B, H, W) or rc < pc.Cout:
            raise ValueError(f"res_up2 shape {tuple(res_up2.shape)} is not the half-resolution of the conv output")
        a.res_up2, a.ldru = res_up2.data_ptr(), rld
    if dbg & 16:                    # developer timeline probe: res2 carries a u64 stamp buffer (see conv3x3_ring.hip)
        if res2 is None or res2.dtype != torch.int64:
            raise ValueError("conv_ring: dbg 16 expects res2 = int64 stamp buffer [workgroups, 8, 4, 10]")
        a.res2, a.ldr2 = res2.data_ptr(), 0
    if out2_cp16 is not None:       # second, fp16 chunk-planar copy of the result (the next Block_'s body[0] source)
        npl = (2 if out2_hl else 1) * (pc.Cout // 16)     # out2_hl: hi | lo planes (the source of a split-fp16 convolution)
        if (out2_cp16.dtype != torch.float16 or tuple(out2_cp16.shape) != (B, npl, H, W, 16)
                or not out2_cp16.is_contiguous()):
            raise ValueError("conv_ring: out2_cp16 must be a contiguous fp16 [B,Cout/16 (x2 with out2_hl),H,W,16] tensor")
        a.out2_cp16, a.out2_lo = out2_cp16.data_ptr(), int(out2_hl)
    check(_lib.lib().cdfo_conv3x3_ring(C.byref(a), _stream()), "cdfo_conv3x3_ring")
    return out


# ----------------------------------------------------------------------------------------------- pointwise


def swap_outer(x: torch.Tensor, B: int, N: int) -> torch.Tensor:
    """x: dense [B*N, ...] -> dense [N*B, ...] (out[n*B+b] = in[b*N+n])."""
    x = x if x.is_contiguous() else x.contiguous()
    out = torch.empty_like(x)
    block = x.numel() // (B * N)
    check(_lib.lib().cdfo_swap_outer(_vp(x), _vp(out), B, N, C.c_longlong(block), _stream()), "cdfo_swap_outer")
    return out


def stem_conv(img: torch.Tensor, img_bstride: int, B: int, H: int, W: int, w: torch.Tensor, bias: torch.Tensor,
              act: int = ACT_NONE, add: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
              out2: Optional[torch.Tensor] = None):
    """img: any fp32 device tensor whose element [b][y][x] sits at data_ptr + (b*img_bstride + y*W + x)*4.
    out / out2: optional dense [B,H,W,64] destinations (slices of a larger batch)."""
    if out is None:
        out = empty_act(B, H, W, 64, img.device)
    elif tuple(out.shape) != (B, H, W, 64) or not out.is_contiguous():
        raise ValueError("stem_conv: out must be a dense [B,H,W,64] tensor")
    lda = ldo2 = 0
    if add is not None:
        _, _, _, _, lda = _chk_act(add, "add")
        if out2 is None:
            out2 = empty_act(B, H, W, 64, img.device)
        elif tuple(out2.shape) != (B, H, W, 64) or not out2.is_contiguous():
            raise ValueError("stem_conv: out2 must be a dense [B,H,W,64] tensor")
        ldo2 = 64
    else:
        out2 = None
    check(_lib.lib().cdfo_stem_conv(_vp(img), C.c_longlong(img_bstride), _vp(w), _vp(bias), B, H, W, act, _vp(out), 64,
                                    _vp(add), lda, _vp(out2), ldo2, _stream()), "cdfo_stem_conv")
    return (out, out2) if add is not None else out


def stem_conv2(img: torch.Tensor, img_bstride: int, B: int, H: int, W: int, wA: torch.Tensor, bA: torch.Tensor,
               add: torch.Tensor, outA: torch.Tensor, wB: torch.Tensor, bB: torch.Tensor, actB: int, outB: torch.Tensor,
               s2dB: bool = False):
    """outA = conv3x3(img; wA, bA) + add, outB = actB(conv3x3(img; wB, bB)); img as in stem_conv, outA dense [B,H,W,64]; outB
    dense [B,H,W,64], or with s2dB a dense [B,H/2+1,W/2+1,256] space-to-depth tensor whose last row / column the caller zeroes."""
    _, _, _, _, lda = _chk_act(add, "add")
    shapeB = (B, H // 2 + 1, W // 2 + 1, 256) if s2dB else (B, H, W, 64)
    if tuple(outA.shape) != (B, H, W, 64) or not outA.is_contiguous() or tuple(outB.shape) != shapeB or not outB.is_contiguous():
        raise ValueError("stem_conv2: outputs must be dense tensors of the documented shapes")
    check(_lib.lib().cdfo_stem_conv2(_vp(img), C.c_longlong(img_bstride), _vp(wA), _vp(bA), _vp(add), lda, _vp(outA), 64,
                                     _vp(wB), _vp(bB), actB, _vp(outB), shapeB[3], int(s2dB), B, H, W, _stream()),
          "cdfo_stem_conv2")


def compose_stem_1x1(w_stem: torch.Tensor, b_stem: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor):
    """A 1x1 convolution (w1 [Co,64,1,1], b1) applied to a one-channel 3x3 stem (w_stem [64,1,3,3], b_stem) with nothing in between
    = one 1 -> Co stencil for the stem kernels (composed in float64): returns (weight [Co,1,3,3], bias [Co]).  Used for
    conv_du_re.0 o conv_expand_rms (arch.py:2200 / 2832, 4446-4449), so that `rms_prior` is never read back."""
    m = w1.detach().double()[:, :, 0, 0]
    return (torch.einsum("oc,ckyx->okyx", m, w_stem.detach().double()).float().contiguous(),
            (m @ b_stem.detach().double() + b1.detach().double()).float().contiguous())


def pack_conv_s2p2_s2d(w2: torch.Tensor, b2: torch.Tensor) -> PackedConv:
    """A 3x3 / stride 2 / pad 2 convolution (64 -> 64) over the space-to-depth form of its input [H/2+1, W/2+1, 4*64] (one zero row /
    column at the end): output (i, j) reads input rows 2i-2+dy = s2d row i-1 phase dy (dy = 0, 1) or s2d row i phase 0 (dy = 2) -- a
    stride-1 pad-1 convolution with weights on the taps (-1, 0) x (-1, 0) only, which the 16-bit MFMA kernel runs with a per-chunk
    tap mask (the exact-fp32 strided kernel it replaces: 0.91 ms per launch at 24 frames)."""
    w2 = w2.detach()
    ws2d = w2.new_zeros(64, 4, 64, 3, 3)
    masks = []
    for a_ in range(2):
        for b_ in range(2):
            m = 0
            for ty in range(2):
                for tx in range(2):
                    dy, dx = 2 * ty + a_, 2 * tx + b_
                    if dy <= 2 and dx <= 2:
                        ws2d[:, a_ * 2 + b_, :, ty, tx] = w2[:, :, dy, dx]
                        m |= 1 << (ty * 3 + tx)
            masks += [m] * 4
    pc2 = pack_conv(ws2d.view(64, 256, 3, 3).contiguous(), b2)
    pc2.tap_mask = torch.tensor(masks, dtype=torch.int32, device=w2.device)
    return pc2


def pack_udsa_head(w0: torch.Tensor, b0: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor):
    """Composition of conv_second (1 -> 64, 3x3, w2 / b2) and the prior U-net's body.0 (64 -> 16, 3x3, w0 / b0) for
    cdfo_udsa_head: (wc [9,9,16], bt [9,16], b0 [16]), summed over the 64 channels in float64."""
    W0 = w0.detach().double().reshape(16, 64, 9)
    W2 = w2.detach().double().reshape(64, 9)
    wc = torch.einsum("oct,cu->tuo", W0, W2).float().contiguous()
    bt = torch.einsum("oct,c->to", W0, b2.detach().double()).float().contiguous()
    return wc, bt, b0.detach().float().contiguous()


def udsa_head(img: torch.Tensor, img_bstride: int, B: int, H: int, W: int, packed) -> torch.Tensor:
    """lrelu(body.0(conv_second(img))) -> [B,H,W,16]; img as in stem_conv."""
    out = empty_act(B, H, W, 16, img.device)
    check(_lib.lib().cdfo_udsa_head(_vp(img), C.c_longlong(img_bstride), _vp(packed[0]), _vp(packed[1]), _vp(packed[2]), B, H, W,
                                    _vp(out), 16, _stream()), "cdfo_udsa_head")
    return out


def layernorm64(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor) -> torch.Tensor:
    B, H, W, Cc, ld = _chk_act(x)
    assert Cc == 64
    out = empty_act(B, H, W, 64, x.device)
    check(_lib.lib().cdfo_layernorm64(_vp(x), ld, _vp(gamma), _vp(beta), C.c_longlong(B * H * W), _vp(out), 64,
                                      _stream()), "cdfo_layernorm64")
    return out


def pack_block_prologue(w_up: torch.Tensor, b_up: torch.Tensor, w_dn: torch.Tensor, b_dn: torch.Tensor):
    """Split-bf16 packing [hi|lo][4][2][128][8] of Block_'s two 1x1 convs (rows 0-63 up.0, 64-127 down.0) + bias[128]."""
    w = torch.cat([w_up.detach().float().reshape(64, 64), w_dn.detach().float().reshape(64, 64)], 0)
    hi = w.bfloat16()
    lo = (w - hi.float()).bfloat16()
    pack = lambda t: t.view(128, 4, 2, 8).permute(1, 2, 0, 3).contiguous().view(-1)
    return torch.cat([pack(hi), pack(lo)]).contiguous(), torch.cat([b_up.detach().float(), b_dn.detach().float()]).contiguous()


def block_prologue(x: torch.Tensor, packed, want_x16: bool = False, lowres_up: bool = False):
    """(u16, d16): fp16 chunk-planar bilinear_x2(up.0(x)) [B,4,2H,2W,16] and down.0(mean2x2(x)) [B,4,H/2,W/2,16];
    want_x16: also (third) the fp16 chunk-planar copy [B,4,H,W,16] of x itself, from the same read of x.
    lowres_up: the first result is up.0(x) at the block's OWN resolution [B,4,H,W,16] (not resampled): the source of
    conv3x3_wino(..., up2=True), which interpolates it on the fly."""
    B, H, W, Cc, ld = _chk_act(x)
    assert Cc == 64 and H % 2 == 0 and W % 2 == 0
    u16 = torch.empty((B, 4, H, W, 16) if lowres_up else (B, 4, 2 * H, 2 * W, 16), dtype=torch.float16, device=x.device)
    d16 = torch.empty((B, 4, H // 2, W // 2, 16), dtype=torch.float16, device=x.device)
    x16 = torch.empty((B, 4, H, W, 16), dtype=torch.float16, device=x.device) if want_x16 else None
    check(_lib.lib().cdfo_block_prologue2(_vp(x), ld, B, H, W, _vp(packed[0]), _vp(packed[1]), _vp(None if lowres_up else u16),
                                          _vp(u16 if lowres_up else None), _vp(d16), _vp(x16), _stream()), "cdfo_block_prologue2")
    return (u16, d16, x16) if want_x16 else (u16, d16)


def pack_qkv_dw(w_qkv: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor):
    """LayerNorm's affine folded into the 1x1 qkv weights: (split-bf16 packing [hi|lo][4][2][192][8], bias W @ beta)."""
    w = w_qkv.detach().float().reshape(192, 64)
    wg = w * gamma.detach().float()[None, :]
    hi = wg.bfloat16()
    lo = (wg - hi.float()).bfloat16()
    pack = lambda t: t.view(192, 4, 2, 8).permute(1, 2, 0, 3).contiguous().view(-1)
    return torch.cat([pack(hi), pack(lo)]).contiguous(), (w @ beta.detach().float()).contiguous()


def qkv_dw(x: torch.Tensor, packed, dw_w: torch.Tensor, eps: float = 1e-5, gram: bool = False):
    """depthwise3x3(conv1x1_64->192(LayerNorm64(x))) in one kernel; packed = pack_qkv_dw(...).
    gram=True: the attention's Gram pass is fused -- returns (v [B,H,W,64], partial [B,n,640], n) for mdta_fold(partial, n, ..)
    and q / k never reach HBM."""
    B, H, W, Cc, ld = _chk_act(x)
    assert Cc == 64
    if gram:
        out = empty_act(B, H, W, 64, x.device)
        n = int(_lib.lib().cdfo_qkv_dw_gram_slots(B, H, W))
        if n < 1:
            raise CdfoError("cdfo_qkv_dw_gram_slots failed")
        part = torch.zeros((B, n, 640), dtype=torch.float32, device=x.device)
        check(_lib.lib().cdfo_qkv_dw(_vp(x), ld, B, H, W, _vp(packed[0]), _vp(packed[1]), _vp(dw_w), C.c_float(eps), _vp(out),
                                     64, _vp(part), n, _stream()), "cdfo_qkv_dw")
        return out, part, n
    out = empty_act(B, H, W, 192, x.device)
    check(_lib.lib().cdfo_qkv_dw(_vp(x), ld, B, H, W, _vp(packed[0]), _vp(packed[1]), _vp(dw_w), C.c_float(eps), _vp(out),
                                 192, None, 0, _stream()), "cdfo_qkv_dw")
    return out


def layernorm64_hl(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor) -> torch.Tensor:
    """LayerNorm64 written as fp16 hi | lo chunk-planar planes [B, 8, H, W, 16] (planes 0-3 hi, 4-7 lo)."""
    B, H, W, Cc, ld = _chk_act(x)
    assert Cc == 64
    out = torch.empty((B, 8, H, W, 16), dtype=torch.float16, device=x.device)
    check(_lib.lib().cdfo_layernorm64_cp16hl(_vp(x), ld, _vp(gamma), _vp(beta), B, C.c_longlong(H * W), _vp(out), _stream()),
          "cdfo_layernorm64_cp16hl")
    return out


def pack_conv_hilo(weight: torch.Tensor, bias: Optional[torch.Tensor], weight_lo: bool = True) -> PackedConv:
    """3x3 weight for the split-fp16 product on the ring kernel: K-expanded (w_hi | w_hi | w_lo) along the input channels,
    to be used with a source of hi | lo planes and plane_wrap = 2 * Cin / 16: a_hi*w_hi + a_lo*w_hi + a_hi*w_lo.
    weight_lo=False drops the third term (weights rounded once to fp16, activations still hi + lo: the "fp16x2" arithmetic
    of the tiled kernel): (w_hi | w_hi), a third fewer K chunks."""
    w = weight.detach().float()
    wh = w.half().float()
    return pack_conv(torch.cat([wh, wh, w - wh] if weight_lo else [wh, wh], 1).contiguous(), bias)


def dwconv3x3(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    B, H, W, Cc, ld = _chk_act(x)
    out = empty_act(B, H, W, Cc, x.device)
    check(_lib.lib().cdfo_dwconv3x3(_vp(x), ld, _vp(w), B, H, W, Cc, _vp(out), Cc, _stream()), "cdfo_dwconv3x3")
    return out


def flow_warp(x: torch.Tensor, mv: torch.Tensor, mv_bstride: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    B, H, W, Cc, ld = _chk_act(x)
    if out is None:
        out = empty_act(B, H, W, Cc, x.device)
    elif tuple(out.shape) != (B, H, W, Cc) or not out.is_contiguous():
        raise ValueError("flow_warp: out must be a dense tensor of the input's shape")
    check(_lib.lib().cdfo_flow_warp(_vp(x), ld, _vp(mv), C.c_longlong(mv_bstride), B, H, W, Cc, _vp(out), Cc,
                                    _stream()), "cdfo_flow_warp")
    return out


def resample2(x: torch.Tensor, up: bool, out: Optional[torch.Tensor] = None, accumulate: bool = False,
              out_f16: bool = False, cp16: bool = False) -> torch.Tensor:
    """cp16 (up only): the result is the fp16 chunk-planar tensor [B, C/16, 2H, 2W, 16] that conv3x3_ws reads."""
    B, H, W, Cc, ld = _chk_act(x)
    Ho, Wo = (2 * H, 2 * W) if up else (H // 2, W // 2)
    if cp16:
        assert up and out is None and not accumulate
        out = torch.empty((B, Cc // 16, Ho, Wo, 16), dtype=torch.float16, device=x.device)
        check(_lib.lib().cdfo_resample2(_vp(x), ld, B, H, W, Cc, _vp(out), 16, 1, 0, 2, _stream()), "cdfo_resample2")
        return out
    odt = torch.float16 if out_f16 else torch.float32
    if out is None:
        assert not accumulate
        out = torch.empty((B, Ho, Wo, Cc), dtype=odt, device=x.device)
    ob, oh, ow, oc, ldo = _chk_act(out, "out", odt)
    assert (ob, oh, ow, oc) == (B, Ho, Wo, Cc)
    check(_lib.lib().cdfo_resample2(_vp(x), ld, B, H, W, Cc, _vp(out), ldo, int(up), int(accumulate), int(out_f16),
                                    _stream()), "cdfo_resample2")
    return out


def scale_channels(x: torch.Tensor, gate: torch.Tensor, want_cp16: bool = False):
    """x * gate[b][c]; want_cp16: also the fp16 chunk-planar copy [B,C/16,H,W,16] of the result (returns a pair)."""
    B, H, W, Cc, ld = _chk_act(x)
    out = empty_act(B, H, W, Cc, x.device)
    o16 = torch.empty((B, Cc // 16, H, W, 16), dtype=torch.float16, device=x.device) if want_cp16 else None
    check(_lib.lib().cdfo_scale_channels(_vp(x), ld, _vp(gate), B, C.c_longlong(H * W), Cc, _vp(out), Cc, _vp(o16),
                                         _stream()), "cdfo_scale_channels")
    return (out, o16) if want_cp16 else out


def upconv_last(x: torch.Tensor, pc: PackedConv, w_last: torch.Tensor, b_last: torch.Tensor, xc: torch.Tensor,
                xc_bstride: int) -> torch.Tensor:
    """Upsampler tail (arch.py:4474-4480) without the HR feature map: lrelu(pixel_shuffle(upconv2(x))) -> conv_last ->
    + bilinear_x4(x_center).  x: [B,2H,2W,64]; pc: upconv2 packed with shuffle2=True; returns [B,1,4H,4W]."""
    B, H2, W2, Cc, ld = _chk_act(x)
    if not pc.shuffle2 or pc.Cout != 256 or pc.Cin != Cc or pc.ks != 1:
        raise ValueError("upconv_last: needs the pixel-shuffle packing of a 1x1 conv with 256 outputs")
    taps = torch.empty((B, 2 * H2, 2 * W2, 12), dtype=torch.float32, device=x.device)
    a = ConvArgs()
    a.src[0], a.ld[0], a.cs[0], a.nsrc = x.data_ptr(), ld, Cc, 1
    a.B, a.H, a.W, a.Ho, a.Wo = B, H2, W2, H2, W2
    a.ks, a.stride, a.pad = 1, 1, 0
    a.Cin, a.Cout, a.CoutP = pc.Cin, pc.Cout, pc.CoutP
    a.w, a.bias, a.act = pc.w.data_ptr(), _p(pc.bias), ACT_LRELU
    a.res2 = w_last.detach().contiguous().float().data_ptr()
    a.out, a.ldo, a.store_mode = taps.data_ptr(), 12, 3
    check(_lib.lib().cdfo_conv1x1_bf16x3(C.byref(a), _stream()), "cdfo_conv1x1_bf16x3 (tap sums)")
    out = torch.empty((B, 1, 2 * H2, 2 * W2), dtype=torch.float32, device=x.device)
    check(_lib.lib().cdfo_conv_last_taps(_vp(taps), 12, _vp(b_last), _vp(xc), C.c_longlong(xc_bstride), B, 2 * H2, 2 * W2,
                                         _vp(out), _stream()), "cdfo_conv_last_taps")
    return out


def conv_last(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, xc: torch.Tensor, xc_bstride: int) -> torch.Tensor:
    B, Hh, Wh, Cc, ld = _chk_act(x)
    assert Cc == 64
    out = torch.empty((B, 1, Hh, Wh), dtype=torch.float32, device=x.device)
    check(_lib.lib().cdfo_conv_last(_vp(x), ld, _vp(w), _vp(bias), _vp(xc), C.c_longlong(xc_bstride), B, Hh, Wh,
                                    _vp(out), _stream()), "cdfo_conv_last")
    return out


def small_conv16(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, stride: int, pad: int, out_pad: int = 0,
                 transposed: bool = False, act: int = ACT_NONE, out_hl: bool = False) -> torch.Tensor:
    """out_hl: return the result as fp16 hi | lo chunk-planar planes [B, 2, Ho, Wo, 16] (source of conv_ring with plane_wrap=2)."""
    B, H, W, Cc, ld = _chk_act(x)
    assert Cc == 16
    if transposed:
        Ho, Wo = (H - 1) * stride - 2 * pad + 3 + out_pad, (W - 1) * stride - 2 * pad + 3 + out_pad
    else:
        Ho, Wo = (H + 2 * pad - 3) // stride + 1, (W + 2 * pad - 3) // stride + 1
    if out_hl:
        hl = torch.empty((B, 2, Ho, Wo, 16), dtype=torch.float16, device=x.device)
        check(_lib.lib().cdfo_small_conv16_hl(_vp(x), ld, _vp(w), _vp(bias), B, H, W, stride, pad, out_pad, int(transposed), act,
                                              _vp(hl), _stream()), "cdfo_small_conv16_hl")
        return hl
    out = empty_act(B, Ho, Wo, 16, x.device)
    check(_lib.lib().cdfo_small_conv16(_vp(x), ld, _vp(w), _vp(bias), B, H, W, stride, pad, out_pad, int(transposed),
                                       act, _vp(out), 16, _stream()), "cdfo_small_conv16")
    return out


def spatial_gate16(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    B, H, W, Cc, ld = _chk_act(x)
    assert Cc == 16
    out = empty_act(B, H, W, 16, x.device)
    check(_lib.lib().cdfo_spatial_gate16(_vp(x), ld, _vp(w), _vp(bias), B, H, W, _vp(out), 16, _stream()),
          "cdfo_spatial_gate16")
    return out


# ----------------------------------------------------------------------------------------------- CVSR_V7 operators
def chan_pool(x: torch.Tensor) -> torch.Tensor:
    """[B,H,W,64] -> [B,H,W,2] = (max over channels, mean over channels)."""
    B, H, W, Cc, ld = _chk_act(x)
    out = torch.empty((B, H, W, 2), dtype=torch.float32, device=x.device)
    check(_lib.lib().cdfo_chan_pool(_vp(x), ld, C.c_longlong(B * H * W), Cc, _vp(out), _stream()), "cdfo_chan_pool")
    return out


def gate_map_cumulative(pooled: torch.Tensor, cum: Optional[torch.Tensor], w: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """G_k = G_{k-1} * sigmoid(conv(G_{k-1} * pooled_0) + bias): the running per-pixel product of a SpatialAttention applied repeatedly
    to the same tensor (x_k = x_0 * G_k); pooled = chan_pool(x_0) [B,H,W,2], cum = G_{k-1} [B,H,W] or None."""
    B, H, W, two = pooled.shape
    if two != 2 or not pooled.is_contiguous() or (cum is not None and (tuple(cum.shape) != (B, H, W) or not cum.is_contiguous())):
        raise ValueError("gate_map_cumulative: pooled [B,H,W,2] and cum [B,H,W] expected")
    out = torch.empty((B, H, W), dtype=torch.float32, device=pooled.device)
    check(_lib.lib().cdfo_gate_map_cumulative(_vp(pooled), _vp(cum), _vp(w), _vp(bias), B, H, W, int(w.shape[-1]), _vp(out), _stream()),
          "cdfo_gate_map_cumulative")
    return out


def spatial_gate(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """SpatialAttention: x * sigmoid(conv(pool(x))); w: the [1,2,ks,ks] weight."""
    B, H, W, Cc, ld = _chk_act(x)
    out = empty_act(B, H, W, Cc, x.device)
    gate = torch.empty((B, H, W), dtype=torch.float32, device=x.device)
    check(_lib.lib().cdfo_spatial_gate(_vp(x), ld, _vp(chan_pool(x)), _vp(w), _vp(bias), B, H, W, Cc, int(w.shape[-1]),
                                       _vp(gate), _vp(out), Cc, _stream()), "cdfo_spatial_gate")
    return out


def rdab_mix(xf: torch.Tensor, xc: torch.Tensor, w3: torch.Tensor, b3: torch.Tensor, vmax: torch.Tensor,
             noise: torch.Tensor) -> torch.Tensor:
    B, H, W, Cc, ld = _chk_act(xf)
    if tuple(noise.shape) != (B, 64, H, W) or not noise.is_contiguous() or noise.dtype != torch.float32:
        raise ValueError(f"rdab_mix: noise must be a contiguous fp32 [B,64,H,W] tensor, got {tuple(noise.shape)}")
    if tuple(vmax.shape) != (B, 64) or tuple(w3.shape) != (1, 2, 3, 3):
        raise ValueError("rdab_mix: vmax must be [B,64] and the spatial weight [1,2,3,3]")
    out = empty_act(B, H, W, 64, xf.device)
    check(_lib.lib().cdfo_rdab_mix(_vp(xf), ld, _vp(chan_pool(xc)), _vp(w3), _vp(b3), _vp(vmax), _vp(noise), B, H, W,
                                   _vp(out), 64, _stream()), "cdfo_rdab_mix")
    return out


def shrink_planes(t: torch.Tensor, level: int) -> torch.Tensor:
    """t: fp32 [B,C,H,W] whose images are dense (any batch stride) -> contiguous [B,C,H>>level,W>>level]."""
    B, Cc, H, W = t.shape
    if t.stride(3) != 1 or t.stride(2) != W or t.stride(1) != H * W:
        t = t.contiguous()
    out = torch.empty((B, Cc, H >> level, W >> level), dtype=torch.float32, device=t.device)
    check(_lib.lib().cdfo_shrink_planes(_vp(t), C.c_longlong(t.stride(0)), Cc, B, H, W, level, _vp(out), _stream()),
          "cdfo_shrink_planes")
    return out


def lincomb(a: torch.Tensor, ca: float, b: Optional[torch.Tensor] = None, cb: float = 0.0,
            c: Optional[torch.Tensor] = None, cc: float = 0.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    for t in (a, b, c, out):
        if t is not None and (not t.is_contiguous() or t.dtype != torch.float32 or t.shape != a.shape):
            raise ValueError("lincomb: contiguous fp32 tensors of one shape expected")
    if out is None:
        out = torch.empty_like(a)
    check(_lib.lib().cdfo_lincomb(_vp(out), _vp(a), float(ca), _vp(b), float(cb), _vp(c), float(cc),
                                  C.c_longlong(a.numel()), _stream()), "cdfo_lincomb")
    return out


# ----------------------------------------------------------------------------------------------- reductions / folds
def nchunks_for(P: int) -> int:
    return max(1, min(128, P // 1024))


def chan_sum_partial(x: torch.Tensor):
    B, H, W, Cc, ld = _chk_act(x)
    assert Cc == 64
    n = nchunks_for(H * W)
    part = torch.empty((B, n, 64), dtype=torch.float32, device=x.device)
    check(_lib.lib().cdfo_chan_sum_partial(_vp(x), ld, B, C.c_longlong(H * W), n, _vp(part), _stream()),
          "cdfo_chan_sum_partial")
    return part, n


def gram_partial(q: torch.Tensor, k: torch.Tensor, ch_per_head: int):
    B, H, W, Cc, ldq = _chk_act(q, "q")
    _, _, _, _, ldk = _chk_act(k, "k")
    n = nchunks_for(H * W)
    part = torch.empty((B, n, 64 * (ch_per_head + 2)), dtype=torch.float32, device=q.device)
    check(_lib.lib().cdfo_gram_partial(_vp(q), ldq, _vp(k), ldk, B, C.c_longlong(H * W), ch_per_head, n, _vp(part),
                                       _stream()), "cdfo_gram_partial")
    return part, n


def mdta_fold(part: torch.Tensor, n: int, temperature: torch.Tensor, proj_w: torch.Tensor) -> PackedConv:
    B = part.shape[0]
    wout = torch.empty((B, 4096), dtype=torch.float32, device=part.device)
    check(_lib.lib().cdfo_mdta_fold(_vp(part), n, _vp(temperature), _vp(proj_w), B, _vp(wout), _stream()),
          "cdfo_mdta_fold")
    return PackedConv(wout, None, 64, 64, 1, 64, False, 4096)


_FOLD_CIN = {}


def fold_scale_inputs(fold: PackedConv, gate: torch.Tensor) -> PackedConv:
    """The per-image 64 x 64 matrices of mdta_fold with a per-image input-channel gate folded in: M'[b] = M[b] diag(gate[b]), so that
    conv(x * gate, M) == conv(x, M') and the gated copy of x is never written (parameter-sized torch arithmetic on [B, 4096])."""
    if fold.Cin != 64 or fold.Cout != 64 or fold.ks != 1 or fold.w_bstride != 4096 or tuple(gate.shape) != (fold.w.shape[0], 64):
        raise ValueError("fold_scale_inputs: a mdta_fold result and a [B, 64] gate expected")
    cin = _FOLD_CIN.get(gate.device)
    if cin is None:       # packed layout [Cin/16][4][CoutP = 64][4]: input channel of every element
        i = torch.arange(4096, device=gate.device)
        cin = (i // 1024) * 16 + ((i // 256) % 4) * 4 + i % 4
        # cached only outside stream capture: a table built while capturing lives in the graph's private pool and is merely RECORDED,
        # so a later eager call would read uninitialised indices
        if not (gate.is_cuda and torch.cuda.is_current_stream_capturing()):
            _FOLD_CIN[gate.device] = cin
    return PackedConv(fold.w * gate[:, cin], None, 64, 64, 1, 64, False, 4096)


def align_fold(gpart, ng, sw, sp, ns, P, temperature, du0_w, du0_b, du2_w, du2_b, proj_w, fusion_w) -> PackedConv:
    B = gpart.shape[0]
    wout = torch.empty((B, 192 * 64), dtype=torch.float32, device=gpart.device)
    check(_lib.lib().cdfo_align_fold(_vp(gpart), ng, _vp(sw), _vp(sp), ns, C.c_longlong(P), _vp(temperature),
                                     _vp(du0_w), _vp(du0_b), _vp(du2_w), _vp(du2_b), _vp(proj_w), _vp(fusion_w), B,
                                     _vp(wout), _stream()), "cdfo_align_fold")
    return PackedConv(wout, None, 64, 192, 1, 64, False, 192 * 64)


def vec_mlp(part: torch.Tensor, n: int, P: int, w1, b1, c1: int, act1: int, w2=None, b2=None, c2: int = 0,
            act2: int = ACT_NONE) -> torch.Tensor:
    B = part.shape[0]
    out = torch.empty((B, c2 if w2 is not None else c1), dtype=torch.float32, device=part.device)
    check(_lib.lib().cdfo_vec_mlp(_vp(part), n, C.c_longlong(P), _vp(w1), _vp(b1), c1, act1, _vp(w2), _vp(b2), c2, act2,
                                  B, _vp(out), _stream()), "cdfo_vec_mlp")
    return out


# ----------------------------------------------------------------------------------------------- prior-fusion attention
def _prep_outs(outs, B, H, W, device):
    if outs is None:
        return empty_act(B, H, W, 64, device), empty_act(B, H, W, 64, device), empty_act(B, H, W, 64, device)
    for t in outs:
        if tuple(t.shape) != (B, H, W, 64) or not t.is_contiguous() or t.dtype != torch.float32:
            raise ValueError("rdab_prep: outs must be three dense fp32 [B,H,W,64] tensors")
    return outs


def rdab_prep(xq: torch.Tensor, vmax: torch.Tensor, noise: torch.Tensor, wW: torch.Tensor, bW: torch.Tensor, outs=None):
    B, H, W, Cc, ld = _chk_act(xq)
    assert Cc == 128 and noise.is_contiguous() and tuple(noise.shape) == (B, 64, H, W)
    sq, vrow, qwin = _prep_outs(outs, B, H, W, xq.device)
    check(_lib.lib().cdfo_rdab_prep(_vp(xq), ld, _vp(vmax), _vp(noise), _vp(wW), _vp(bW), B, C.c_longlong(H * W),
                                    _vp(sq), 64, _vp(vrow), 64, _vp(qwin), 64, _stream()), "cdfo_rdab_prep")
    return sq, vrow, qwin


def rdab_prep_rng(xq: torch.Tensor, vmax: torch.Tensor, seed: int, draw: int, wW: torch.Tensor, bW: torch.Tensor,
                  noise_out: Optional[torch.Tensor] = None, outs=None):
    """rdab_prep with the uniform draws of arch.py:2169 generated inside the kernel (Philox4x32-10, key = seed, `draw` =
    index of the call within the forward).  noise_out: optional fp32 [B,64,H,W] tensor that receives the drawn values."""
    B, H, W, Cc, ld = _chk_act(xq)
    assert Cc == 128
    if noise_out is not None and (tuple(noise_out.shape) != (B, 64, H, W) or not noise_out.is_contiguous()
                                  or noise_out.dtype != torch.float32):
        raise ValueError("rdab_prep_rng: noise_out must be a contiguous fp32 [B,64,H,W] tensor")
    sq, vrow, qwin = _prep_outs(outs, B, H, W, xq.device)
    if isinstance(seed, torch.Tensor):      # the key lives in device memory (int64[1]): graph replays re-read it
        if seed.dtype != torch.int64 or seed.numel() != 1 or seed.device != xq.device:
            raise ValueError("rdab_prep_rng: a device-side key must be an int64[1] tensor on the operands' device")
        check(_lib.lib().cdfo_rdab_prep_rng_dev(_vp(xq), ld, _vp(vmax), _vp(seed), draw, _vp(noise_out), _vp(wW), _vp(bW), B,
                                                C.c_longlong(H * W), _vp(sq), 64, _vp(vrow), 64, _vp(qwin), 64, _stream()),
              "cdfo_rdab_prep_rng_dev")
        return sq, vrow, qwin
    check(_lib.lib().cdfo_rdab_prep_rng(_vp(xq), ld, _vp(vmax), C.c_longlong(seed & 0x7FFFFFFFFFFFFFFF), draw, _vp(noise_out),
                                        _vp(wW), _vp(bW), B, C.c_longlong(H * W), _vp(sq), 64, _vp(vrow), 64, _vp(qwin), 64,
                                        _stream()), "cdfo_rdab_prep_rng")
    return sq, vrow, qwin


_seed_counter = 0


def next_noise_seed(device) -> int:
    """A fresh 63-bit Philox key per forward, reproducible under torch.manual_seed: taken from (and advancing) the state
    of torch's default generator of `device` -- plumbing only, no random numbers are drawn by torch."""
    global _seed_counter
    if torch.cuda.is_current_stream_capturing():
        # the generator refuses state changes during a graph capture; a captured forward reads its key from device memory
        # anyway (CVSR_V8.refresh_noise_key rewrites it before every replay), so this value only seeds the capture run
        base, off = int(torch.initial_seed()), 4 * _seed_counter
        _seed_counter += 1
    else:
        g = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
        base, off = int(g.initial_seed()), int(g.get_offset())
        g.set_offset(off + 4)
    return (base * 0x9E3779B97F4A7C15 + off * 0xD1B54A32D192ED03 + 0x632BE59BD9B4E019) & 0x7FFFFFFFFFFFFFFF


def colconv9(x: torch.Tensor, wH: torch.Tensor, bH: torch.Tensor) -> torch.Tensor:
    B, H, W, Cc, ld = _chk_act(x)
    out = empty_act(B, H, W, 64, x.device)
    check(_lib.lib().cdfo_colconv9(_vp(x), ld, _vp(wH), _vp(bH), B, H, W, _vp(out), 64, _stream()), "cdfo_colconv9")
    return out


def seq_attn(q: torch.Tensor, v: torch.Tensor, mode: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    B, H, W, Cc, ldq = _chk_act(q, "q")
    _, _, _, _, ldv = _chk_act(v, "v")
    if out is None:
        out = empty_act(B, H, W, 64, q.device)
    _, _, _, _, ldo = _chk_act(out, "out")
    check(_lib.lib().cdfo_seq_attn(_vp(q), ldq, _vp(v), ldv, _vp(out), ldo, B, H, W, mode, _stream()), "cdfo_seq_attn")
    return out
