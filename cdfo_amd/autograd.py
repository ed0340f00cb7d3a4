"""torch.autograd plumbing of the CVSR_V8 training path (SURVEY section 8f n2 / boundary B1: `train_LD_37.py:376-381` calls
``model(...)`` then ``loss.backward()``).

Every operator that touches pixels is a ``torch.autograd.Function`` whose forward AND backward run in libcdfo_hip.so:
the forward through the entry points of the inference path's ``precision="f32"`` / ``"bf16x3"`` modes (which of the two the
convolutions use: ``CONV_PREC`` below; everything that is not a convolution is exact fp32), the backward through
the same entry points with adjoint operands (flipped / transposed weights, swapped roles) or the kernels of
``csrc/train_ops.hip``.  torch itself is used for what it is here for -- the autograd graph, device memory, views /
permutes -- and for arithmetic on *parameter-sized or per-image* tensors only (weight flips, the 8x8 / 16x16 attention
matrices per image and head, the 64-vector gate MLPs): nothing of size O(pixels) is computed by ATen.

Layout: fp32 pixel-major ``[B, H, W, C]`` like the rest of the package.  There is no CPU fallback."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import torch
from torch.autograd import Function

from . import _lib
from . import kernels as K
from ._lib import check
from .kernels import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, PackedConv, _stream, _vp

F32 = K.PREC_F32
# Arithmetic of the training path's convolutions -- `CONV_PREC`, read at every call.
#   PREC_BF16X3 (default since round 3): split-bf16 three-pass MFMA, fp32 accumulate (each operand is carried as bf16 hi + lo,
#     ~16 significant bits; products are fp32-grade, ~1e-6 relative).  It applies to exactly three kernel families:
#       (1) 3x3 / stride 1 / pad 1 convolutions with Cin % 16 == 0 -- forward AND input gradient (the adjoint is such a
#           convolution) -- on the tiled 16-bit kernel (cdfo_conv3x3_bf16);
#       (2) 1x1 convolutions whose sources are multiples of 64 channels and whose padded Cout is a multiple of 64 (<= 256):
#           kernels.conv routes them to cdfo_conv1x1_bf16x3 (qkv, project_out, input_conv, fuse, fusion_out, down.0 / up.0,
#           tsa_fusion, upconv1 / 2 -- forward and input gradient);
#       (3) the weight gradient of 3x3 convolutions with both channel counts >= 16 (cdfo_conv_wgrad_prec, conv_wgrad below).
#     Everything else -- stems, the 16-channel prior U-net, per-image attention applications (prec=F32 below), depthwise,
#     LayerNorm, attention, warp, resampling, every bias / 1x1 / depthwise weight gradient -- is exact fp32.
#     Gradient accuracy against the reference's own gradients: median 1e-5 - 4e-5 of max|g| per tensor, forward `out` <= 3e-5.
#   PREC_F32 (CDFO_TRAIN_EXACT=1 in the environment, or `autograd.CONV_PREC = kernels.PREC_F32`): exact-fp32 MFMA everywhere
#     (median 3e-7 - 1e-6, forward <= 1e-5), at 1/16 of the 16-bit matrix rate: 425 ms instead of 195 ms per step at 20 x 64 x 64.
#   tests/test_gpu_train.py gates BOTH modes, each with its own tolerances.
import os as _os
CONV_PREC = K.PREC_F32 if _os.environ.get("CDFO_TRAIN_EXACT", "0") not in ("", "0") else K.PREC_BF16X3


def conv_precision_name() -> str:
    """What the training path's convolutions compute in right now (reported by bench.py's train_step line)."""
    return "exact fp32 MFMA" if CONV_PREC == K.PREC_F32 else "split-bf16 3-pass MFMA, fp32 accumulate (fp32-grade products)"


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


def _ld(t: torch.Tensor) -> int:
    return t.stride(-2) if t.dim() >= 2 else 1


# ------------------------------------------------------------------------------------------------ low-level wrappers
def ew(a: torch.Tensor, b: Optional[torch.Tensor], mode: int, aux: int = 0, scale: float = 1.0, P: int = 0,
       shape=None) -> torch.Tensor:
    """mode 0 a*b, 1 a*(1-b), 2 a+b, 3 a*act'(y=b) (aux = act), 4 row broadcast of a[nimg, C] * scale over P pixels."""
    if mode == 4:
        nimg, Cc = a.shape
        out = torch.empty(shape, dtype=torch.float32, device=a.device)
        rows = out.numel() // Cc
        check(_lib.lib().cdfo_ew(_vp(_c(a)), Cc, None, 0, C.c_longlong(rows), Cc, 4, 0, float(scale), C.c_longlong(P), _vp(out), Cc,
                                 _stream()), "cdfo_ew")
        return out
    Cc = a.shape[-1]
    out = torch.empty(a.shape, dtype=torch.float32, device=a.device)
    rows = out.numel() // Cc
    check(_lib.lib().cdfo_ew(_vp(a), a.stride(-2), _vp(b), b.stride(-2), C.c_longlong(rows), Cc, mode, aux, float(scale),
                             C.c_longlong(1), _vp(out), Cc, _stream()), "cdfo_ew")
    return out


def _dense_rows(t: torch.Tensor) -> torch.Tensor:
    """[..., C] tensor whose leading dims are densely packed at pitch ld = stride(-2) (channel slices qualify)."""
    if t.stride(-1) != 1:
        return t.contiguous()
    ld = t.stride(-2)
    exp = ld
    for d in range(t.dim() - 2, -1, -1):
        if t.shape[d] != 1 and t.stride(d) != exp:
            return t.contiguous()
        exp *= t.shape[d]
    return t


def act_bwd(g: torch.Tensor, y: torch.Tensor, act: int) -> torch.Tensor:
    if act == ACT_NONE:
        return g
    return ew(_dense_rows(g), _dense_rows(y), 3, act)


def coldot(a: torch.Tensor, b: Optional[torch.Tensor], nimg: int, scale: float = 1.0) -> torch.Tensor:
    """out[img, c] = scale * sum_p a[img, p, c] * (b[img, p, c] | 1); a, b: [..., C] with nimg * P rows."""
    a = _dense_rows(a)
    Cc = a.shape[-1]
    rows = a.numel() // Cc if a.is_contiguous() else a.shape[:-1].numel()
    P = rows // nimg
    nchunk = max(1, min(512, P // 64))      # (round 3: 64 chunks of >= 256 rows left most of the GPU idle: 160 us per bias gradient)
    part = torch.empty(nimg * nchunk * Cc, dtype=torch.float32, device=a.device)
    out = torch.empty((nimg, Cc), dtype=torch.float32, device=a.device)
    if b is not None:
        b = _dense_rows(b)
    check(_lib.lib().cdfo_coldot(_vp(a), a.stride(-2), _vp(b), 0 if b is None else b.stride(-2), nimg, C.c_longlong(P), Cc, nchunk,
                                 float(scale), _vp(part), _vp(out), _stream()), "cdfo_coldot")
    return out


def conv_wgrad(S: torch.Tensor, L: torch.Tensor, ks: int, stride: int, pad: int, dw: torch.Tensor, Btot: int, b_off: int):
    """dw[a][b_off + b][ky][kx] (+)= sum S[n,y,x,a] * L[n, y*s+ky-p, x*s+kx-p, b]  (assigned; see csrc/train_ops.hip)."""
    S, L = _dense_rows(S), _dense_rows(L)
    N, Hs, Ws, A = S.shape
    _, Hl, Wl, Bc = L.shape
    nblk = ((A + 63) // 64) * ((Bc + 63) // 64) * ks * ks
    nsplit = max(1, min(N * Hs // 4 if N * Hs >= 4 else 1, 1024 // nblk if nblk < 1024 else 1, 256))
    n = int(_lib.lib().cdfo_conv_wgrad_slab_floats(A, Bc, ks, nsplit))
    slab = torch.empty(n, dtype=torch.float32, device=S.device)
    # the 16-bit form where the convolutions themselves use it: the 3x3 convolutions (tiled split-bf16 kernel) and, since round 5, the 1x1
    # ones (cdfo_conv1x1_bf16x3 in the forward): their exact-fp32 MFMA weight gradients were 15 ms of a 160 ms backward at 1/16 of the rate
    prec = CONV_PREC if (ks in (1, 3) and stride == 1 and A >= 16 and Bc >= 16) else F32
    check(_lib.lib().cdfo_conv_wgrad_prec(_vp(S), S.stride(-2), A, _vp(L), L.stride(-2), Bc, N, Hs, Ws, Hl, Wl, ks, stride, pad, nsplit,
                                          _vp(slab), _vp(dw), Btot, b_off, prec, _stream()), "cdfo_conv_wgrad_prec")


def _pack_per_image(Wb: torch.Tensor) -> PackedConv:
    """Wb [B, Cout, Cin] (Cout % 32 == 0, Cin % 4 == 0) -> the per-image fp32 packing [B][Cin/4][Cout][4] of conv_igemm."""
    B, Co, Ci = Wb.shape
    w = Wb.view(B, Co, Ci // 4, 4).permute(0, 2, 1, 3).contiguous().view(B, -1)
    return PackedConv(w, None, Co, Ci, 1, Co, False, Ci * Co)


# ------------------------------------------------------------------------------------------------ convolution
class _Conv(Function):
    """y = act(conv(cat(srcs), W, b)) [+ res...]; 3x3 / 1x1, stride 1 (stride 2 only where no input gradient is needed)."""

    @staticmethod
    def forward(ctx, weight, bias, stride, pad, act, nsrc, *ts):
        srcs, res = list(ts[:nsrc]), [t for t in ts[nsrc:] if t is not None]
        pc = K.pack_conv(weight.detach(), None if bias is None else bias.detach())
        srcs_d = [_dense_rows(s.detach()) for s in srcs]
        y = K.conv(srcs_d, pc, stride=stride, pad=pad, act=act, prec=CONV_PREC)
        out = y
        for r in res:
            out = ew(out, _dense_rows(r.detach()), 2)
        ctx.meta = (stride, pad, act, nsrc, len(ts) - nsrc, [s.shape[-1] for s in srcs])
        ctx.save_for_backward(weight, y if act != ACT_NONE else None, *srcs_d)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, g):
        stride, pad, act, nsrc, nres, cs = ctx.meta
        weight, y = ctx.saved_tensors[0], ctx.saved_tensors[1]
        srcs = ctx.saved_tensors[2:]
        g = _c(g)
        gp = act_bwd(g, y, act) if act != ACT_NONE else g
        Co, Ci, ks, _ = weight.shape
        dW = db = None
        if ctx.needs_input_grad[0]:
            dW = torch.empty_like(weight)
            off = 0
            for s, c in zip(srcs, cs):
                conv_wgrad(gp, s, ks, stride, pad, dW, Ci, off)
                off += c
        if ctx.has_bias and ctx.needs_input_grad[1]:
            db = coldot(gp, None, 1).view(-1)
        dsrcs: List[Optional[torch.Tensor]] = [None] * nsrc
        if any(ctx.needs_input_grad[6:6 + nsrc]):
            if stride != 1:
                raise NotImplementedError("input gradient of a strided convolution is only implemented for the 16-channel layers")
            wt = weight.detach().flip(2, 3).transpose(0, 1).contiguous()           # [Ci][Co][k][k]: the adjoint's weights
            dx = K.conv([gp], K.pack_conv(wt, None), stride=1, pad=ks - 1 - pad, prec=CONV_PREC)
            off = 0
            for i, c in enumerate(cs):
                if ctx.needs_input_grad[6 + i]:
                    dsrcs[i] = dx[..., off:off + c] if nsrc > 1 else dx
                off += c
        dres = [g if ctx.needs_input_grad[6 + nsrc + i] else None for i in range(nres)]
        return (dW, db, None, None, None, None, *dsrcs, *dres)


def conv(srcs, weight, bias=None, stride=1, pad=0, act=ACT_NONE, res: Sequence[torch.Tensor] = ()):
    if isinstance(srcs, torch.Tensor):
        srcs = [srcs]
    return _Conv.apply(weight, bias, stride, pad, act, len(srcs), *srcs, *res)


class _Stem(Function):
    """3x3 conv 1 -> 64 on a single-channel plane [N, H, W] (+ LeakyReLU): inputs of the network, so only dW / db."""

    @staticmethod
    def forward(ctx, img, weight, bias, act):
        N, H, W = img.shape
        img = _c(img.detach().float())
        y = K.stem_conv(img, H * W, N, H, W, _c(weight.detach()), _c(bias.detach()), act)
        ctx.save_for_backward(img, y if act != ACT_NONE else None, weight)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, g):
        img, y, weight = ctx.saved_tensors
        gp = act_bwd(_c(g), y, ctx.act) if ctx.act != ACT_NONE else _c(g)
        dW = torch.empty_like(weight)
        conv_wgrad(gp, img.unsqueeze(-1), 3, 1, 1, dW, 1, 0)
        return None, dW, coldot(gp, None, 1).view(-1), None


def stem(img, weight, bias, act=ACT_NONE):
    return _Stem.apply(img, weight, bias, act)


class _ConvLast(Function):
    """conv_last (3x3, 64 -> 1) + bias + bilinear x4 of the centre LR frame (arch.py:4476-4480) -> [B, 1, 4H, 4W]."""

    @staticmethod
    def forward(ctx, t, weight, bias, xc, xc_bstride):
        t = _dense_rows(t.detach())
        ctx.save_for_backward(t, weight)
        return K.conv_last(t, _c(weight.detach()), _c(bias.detach()), xc, xc_bstride)

    @staticmethod
    def backward(ctx, g):
        t, weight = ctx.saved_tensors
        B, Hh, Wh, _ = t.shape
        gpl = _c(g).view(B, Hh, Wh)
        dW = torch.empty_like(weight)
        conv_wgrad(gpl.unsqueeze(-1), t, 3, 1, 1, dW, 64, 0)
        db = coldot(gpl.reshape(B * Hh * Wh, 1), None, 1).view(-1)
        # dt[p][c] = sum_taps w[0][c][2-ky][2-kx] g[p + tap - 1]: a 1 -> 64 stem convolution with flipped taps
        wt = weight.detach().flip(2, 3).transpose(0, 1).contiguous()              # [64][1][3][3]
        dt = K.stem_conv(gpl, Hh * Wh, B, Hh, Wh, wt, torch.zeros(64, device=g.device), ACT_NONE)
        return dt, dW, db, None, None


# ------------------------------------------------------------------------------------------------ normalisation, depthwise
class _LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta):
        x = _dense_rows(x.detach())
        ctx.save_for_backward(x, gamma)
        return K.layernorm64(x, _c(gamma.detach()), _c(beta.detach()))

    @staticmethod
    def backward(ctx, g):
        x, gamma = ctx.saved_tensors
        g = _c(g)
        npix = x.shape[0] * x.shape[1] * x.shape[2]
        nblk = max(1, min(512, npix // 64))
        part = torch.empty((nblk, 2, 64), dtype=torch.float32, device=x.device)
        dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        check(_lib.lib().cdfo_layernorm64_bwd(_vp(x), x.stride(-2), _vp(g), 64, _vp(_c(gamma.detach())), C.c_longlong(npix), _vp(dx), 64,
                                              _vp(part), nblk, _stream()), "cdfo_layernorm64_bwd")
        s = part.sum(0)                                                           # [2, 64]: parameter-sized
        return dx, s[0], s[1]


class _DwConv(Function):
    @staticmethod
    def forward(ctx, x, weight):
        x = _dense_rows(x.detach())
        ctx.save_for_backward(x, weight)
        return K.dwconv3x3(x, _c(weight.detach()))

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = _c(g)
        B, H, W, Cc = x.shape
        dx = K.dwconv3x3(g, weight.detach().flip(2, 3).contiguous())
        nblk = max(1, min(4096, B * H * W // 64))       # (round 3: 256 blocks = 2240 pixels per thread, 5.1 ms per launch)
        part = torch.empty((nblk, 9, Cc), dtype=torch.float32, device=x.device)
        check(_lib.lib().cdfo_dwconv3x3_wgrad(_vp(x), x.stride(-2), _vp(g), Cc, B, H, W, Cc, _vp(part), nblk, _stream()),
              "cdfo_dwconv3x3_wgrad")
        return dx, part.sum(0).t().reshape(Cc, 1, 3, 3)


# ------------------------------------------------------------------------------------------------ prior U-net pieces
class _SmallConv16(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, out_pad, transposed, act):
        x = _dense_rows(x.detach())
        y = K.small_conv16(x, _c(weight.detach()), _c(bias.detach()), stride, pad, out_pad, transposed, act)
        ctx.save_for_backward(x, weight, y if act != ACT_NONE else None)
        ctx.meta = (stride, pad, out_pad, transposed, act)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, y = ctx.saved_tensors
        stride, pad, out_pad, transposed, act = ctx.meta
        gp = act_bwd(_c(g), y, act) if act != ACT_NONE else _c(g)
        H, Ho, W, Wo = x.shape[1], gp.shape[1], x.shape[2], gp.shape[2]
        if not transposed and H - ((Ho - 1) * stride - 2 * pad + 3) != W - ((Wo - 1) * stride - 2 * pad + 3):
            raise NotImplementedError("small_conv16 backward: rows and columns need the same output padding (H, W of equal parity)")
        zero = torch.zeros(16, device=g.device)
        dW = torch.empty_like(weight)
        if not transposed:
            dx = K.small_conv16(gp, _c(weight.detach()), zero, stride, pad, H - ((Ho - 1) * stride - 2 * pad + 3), True)
            conv_wgrad(gp, x, 3, stride, pad, dW, 16, 0)
        else:
            dx = K.small_conv16(gp, _c(weight.detach()), zero, stride, pad, 0, False)
            conv_wgrad(x, gp, 3, stride, pad, dW, 16, 0)
        assert dx.shape == x.shape, (dx.shape, x.shape)
        return dx, dW, coldot(gp, None, 1).view(-1), None, None, None, None, None


class _SpatialGate16(Function):
    @staticmethod
    def forward(ctx, t, weight, bias):
        t = _dense_rows(t.detach())
        ctx.save_for_backward(t, weight, bias)
        return K.spatial_gate16(t, _c(weight.detach()), _c(bias.detach()))

    @staticmethod
    def backward(ctx, g):
        t, weight, bias = ctx.saved_tensors
        g = _c(g)
        B, H, W, _ = t.shape
        scratch = torch.empty(B * H * W * 5, dtype=torch.float32, device=t.device)
        dt = torch.empty(t.shape, dtype=torch.float32, device=t.device)
        dw = torch.empty(99, dtype=torch.float32, device=t.device)
        check(_lib.lib().cdfo_spatial_gate16_bwd(_vp(t), t.stride(-2), _vp(g), 16, _vp(_c(weight.detach())), _vp(_c(bias.detach())), B, H, W,
                                                 _vp(scratch), _vp(dt), 16, _vp(dw), _stream()), "cdfo_spatial_gate16_bwd")
        return dt, dw[:98].view(1, 2, 7, 7), dw[98:99]


# ------------------------------------------------------------------------------------------------ channel attention
def _attn_matrices(part: torch.Tensor, chp: int, temp: torch.Tensor):
    """gram partials [B, n, 64*(chp+2)] -> (A [B,h,chp,chp], Gn, nq [B,64], nk [B,64]); per-image, head-sized tensors."""
    s = part.sum(1).view(part.shape[0], 64, chp + 2)
    B = s.shape[0]
    heads = 64 // chp
    G = s[..., :chp].reshape(B, heads, chp, chp)
    nq = s[..., chp].sqrt().clamp_min(1e-12)                                        # F.normalize: x / max(||x||, eps)
    nk = s[..., chp + 1].sqrt().clamp_min(1e-12)
    Gn = G / (nq.view(B, heads, chp, 1) * nk.view(B, heads, 1, chp))
    A = torch.softmax(Gn * temp.view(1, heads, 1, 1), dim=-1)
    return A, Gn, nq, nk


def _blockdiag(M: torch.Tensor) -> torch.Tensor:
    """[B, heads, c, c] -> [B, 64, 64] block diagonal."""
    B, h, c, _ = M.shape
    out = M.new_zeros(B, h, c, h, c)
    idx = torch.arange(h, device=M.device)
    out[:, idx, :, idx, :] = M.permute(1, 0, 2, 3)
    return out.view(B, h * c, h * c)


class _ChannelAttention(Function):
    """softmax(normalize(q) normalize(k)^T * temperature) v over all pixels, heads x chp channels (arch.py:1555-1571,
    3463-3472).  The products over pixels are Gram / per-image 1x1 kernels; the head-sized matrices are torch tensors."""

    @staticmethod
    def forward(ctx, q, k, v, temp, heads):
        q, k, v = (_dense_rows(t.detach()) for t in (q, k, v))
        chp = 64 // heads
        part, _ = K.gram_partial(q, k, chp)
        A, Gn, nq, nk = _attn_matrices(part, chp, temp.detach().view(-1))
        out = K.conv([v], _pack_per_image(_blockdiag(A)), prec=F32)
        ctx.save_for_backward(q, k, v, temp, A, Gn, nq, nk)
        ctx.chp = chp
        return out

    @staticmethod
    def backward(ctx, g):
        q, k, v, temp, A, Gn, nq, nk = ctx.saved_tensors
        chp = ctx.chp
        heads = 64 // chp
        g = _c(g)
        B = q.shape[0]
        part, _ = K.gram_partial(g, v, chp)                                       # dA[h,i,j] = sum_p g[(h,i)] v[(h,j)]
        dA = part.sum(1).view(B, 64, chp + 2)[..., :chp].reshape(B, heads, chp, chp)
        dv = K.conv([g], _pack_per_image(_blockdiag(A.transpose(-1, -2))), prec=F32)
        dS = A * (dA - (dA * A).sum(-1, keepdim=True))
        t = temp.detach().view(1, heads, 1, 1)
        dtemp = (dS * Gn).sum((0, 2, 3)).view_as(temp)
        dGn = dS * t
        inq, ink = 1.0 / nq.view(B, heads, chp), 1.0 / nk.view(B, heads, chp)
        M = dGn * inq.unsqueeze(-1) * ink.unsqueeze(-2)                            # dGn[i,j] / (nq_i nk_j)
        sq = (dGn * Gn).sum(-1) * inq * inq                                        # row sums / nq_i^2
        sk = (dGn * Gn).sum(-2) * ink * ink
        Wq = torch.cat([_blockdiag(M), torch.diag_embed(-sq.reshape(B, 64))], 2)   # dq = M k - diag(sq) q
        Wk = torch.cat([_blockdiag(M.transpose(-1, -2)), torch.diag_embed(-sk.reshape(B, 64))], 2)
        dq = K.conv([k, q], _pack_per_image(Wq), prec=F32)
        dk = K.conv([q, k], _pack_per_image(Wk), prec=F32)
        return dq, dk, dv, dtemp, None


# ------------------------------------------------------------------------------------------------ gates, means
class _ChanMean(Function):
    @staticmethod
    def forward(ctx, x):
        x = _dense_rows(x.detach())
        ctx.shape = x.shape
        B = x.shape[0]
        return coldot(x, None, B, 1.0 / (x.shape[1] * x.shape[2]))

    @staticmethod
    def backward(ctx, g):
        B, H, W, Cc = ctx.shape
        return ew(_c(g), None, 4, scale=1.0 / (H * W), P=H * W, shape=ctx.shape)


class _ScaleChannels(Function):
    @staticmethod
    def forward(ctx, x, gate):
        x = _dense_rows(x.detach())
        ctx.save_for_backward(x, gate)
        return K.scale_channels(x, _c(gate.detach()))

    @staticmethod
    def backward(ctx, g):
        x, gate = ctx.saved_tensors
        g = _c(g)
        return K.scale_channels(g, _c(gate.detach())), coldot(g, x, x.shape[0])


class _MulMask(Function):
    """x * mask (inv = False) or x * (1 - mask); the hard mask carries no gradient (arch.py:2194-2195: masked_fill).
    `anchor` (optional scalar built from the mask generator's parameters) receives a zero gradient, so that those parameters end
    the backward pass with zero-filled .grad tensors as they do in the reference (not None: Adam's weight decay sees them)."""

    @staticmethod
    def forward(ctx, x, mask, inv, anchor):
        ctx.save_for_backward(mask)
        ctx.inv = inv
        ctx.has_anchor = anchor is not None
        return ew(_dense_rows(x.detach()), mask, 1 if inv else 0)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return ew(_c(g), mask, 1 if ctx.inv else 0), None, None, (g.new_zeros(()) if ctx.has_anchor else None)


def gumbel_mask(vmax: torch.Tensor, noise, B: int, H: int, W: int, capture: Optional[list] = None) -> torch.Tensor:
    """The hard mask of arch.py:2168-2195 as a [B,H,W,64] tensor (no gradient).  noise: a [B,64,H,W] tensor, or
    ("rng", seed, draw) to draw the uniforms in the kernel exactly as the inference path's mask kernel does."""
    mask = torch.empty((B, H, W, 64), dtype=torch.float32, device=vmax.device)
    vmax = _c(vmax.detach())
    if isinstance(noise, tuple):
        _, seed, draw = noise
        cap = None
        if capture is not None:
            cap = torch.empty((B, 64, H, W), dtype=torch.float32, device=vmax.device)
            capture.append(cap)
        check(_lib.lib().cdfo_gumbel_mask(_vp(vmax), None, C.c_longlong(seed & 0x7FFFFFFFFFFFFFFF), draw, _vp(cap), B,
                                          C.c_longlong(H * W), _vp(mask), 64, _stream()), "cdfo_gumbel_mask")
    else:
        check(_lib.lib().cdfo_gumbel_mask(_vp(vmax), _vp(_c(noise.float())), C.c_longlong(0), 0, None, B, C.c_longlong(H * W),
                                          _vp(mask), 64, _stream()), "cdfo_gumbel_mask")
    return mask


# ------------------------------------------------------------------------------------------------ 9-tap convolutions
def _corr9(x, g, axis):
    B, H, W, _ = x.shape
    nblk = max(1, min(256, B * H * W // 64))
    part = torch.empty((nblk, 10), dtype=torch.float32, device=x.device)
    check(_lib.lib().cdfo_corr9(_vp(x), x.stride(-2), _vp(g), g.stride(-2), axis, B, H, W, _vp(part), nblk, _stream()), "cdfo_corr9")
    return part.sum(0)


class _ChanConv9(Function):
    """directW1_conv (arch.py:2161, 2216-2219): 9 taps along the channel axis, one shared weight vector + bias."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = _dense_rows(x.detach())
        ctx.save_for_backward(x, weight)
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        npix = x.shape[0] * x.shape[1] * x.shape[2]
        check(_lib.lib().cdfo_chanconv9(_vp(x), x.stride(-2), _vp(_c(weight.detach())), _vp(_c(bias.detach())), 0, C.c_longlong(npix),
                                        _vp(out), 64, _stream()), "cdfo_chanconv9")
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = _c(g)
        dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        npix = x.shape[0] * x.shape[1] * x.shape[2]
        check(_lib.lib().cdfo_chanconv9(_vp(g), 64, _vp(_c(weight.detach())), None, 1, C.c_longlong(npix), _vp(dx), 64, _stream()),
              "cdfo_chanconv9")
        s = _corr9(x, g, 0)
        return dx, s[:9].view_as(weight), s[9:10]


class _ColConv9(Function):
    """directH1_conv (arch.py:2162, 2225): 9 taps along the image rows."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = _dense_rows(x.detach())
        ctx.save_for_backward(x, weight)
        return K.colconv9(x, _c(weight.detach()), _c(bias.detach()))

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = _c(g)
        dx = K.colconv9(g, weight.detach().flip(2).contiguous(), torch.zeros(1, device=g.device))
        s = _corr9(x, g, 1)
        return dx, s[:9].view_as(weight), s[9:10]


# ------------------------------------------------------------------------------------------------ sequence attention, warp
class _SeqAttn(Function):
    """softmax(Q Q^T) V per row (0) / column (1) / 8x8 window (2) (arch.py:2220-2243)."""

    @staticmethod
    def forward(ctx, q, v, mode):
        q, v = _dense_rows(q.detach()), _dense_rows(v.detach())
        o = K.seq_attn(q, v, mode)
        ctx.save_for_backward(q, v, o)
        ctx.mode = mode
        return o

    @staticmethod
    def backward(ctx, g):
        q, v, o = ctx.saved_tensors
        g = _c(g)
        B, H, W, _ = q.shape
        dq = torch.empty((B, H, W, 64), dtype=torch.float32, device=q.device)
        dv = torch.empty((B, H, W, 64), dtype=torch.float32, device=q.device)
        check(_lib.lib().cdfo_seq_attn_bwd(_vp(q), q.stride(-2), _vp(v), v.stride(-2), _vp(o), 64, _vp(g), 64, _vp(dq), 64, _vp(dv), 64,
                                           B, H, W, ctx.mode, _stream()), "cdfo_seq_attn_bwd")
        return dq, dv, None


class _FlowWarp(Function):
    @staticmethod
    def forward(ctx, x, mv, mv_bstride):
        x = _dense_rows(x.detach())
        ctx.save_for_backward(mv)
        ctx.meta = (x.shape, mv_bstride)
        return K.flow_warp(x, mv, mv_bstride)

    @staticmethod
    def backward(ctx, g):
        (mv,) = ctx.saved_tensors
        (B, H, W, Cc), bs = ctx.meta
        g = _c(g)
        dx = torch.zeros((B, H, W, Cc), dtype=torch.float32, device=g.device)
        check(_lib.lib().cdfo_flow_warp_bwd(_vp(g), Cc, _vp(mv), C.c_longlong(bs), B, H, W, Cc, _vp(dx), Cc, _stream()), "cdfo_flow_warp_bwd")
        return dx, None, None


class _Resample2(Function):
    @staticmethod
    def forward(ctx, x, up):
        x = _dense_rows(x.detach())
        ctx.meta = (x.shape, up)
        return K.resample2(x, up=up)

    @staticmethod
    def backward(ctx, g):
        (B, H, W, Cc), up = ctx.meta
        g = _c(g)
        din = torch.empty((B, H, W, Cc), dtype=torch.float32, device=g.device)
        check(_lib.lib().cdfo_resample2_bwd(_vp(g), Cc, B, H, W, Cc, int(up), _vp(din), Cc, _stream()), "cdfo_resample2_bwd")
        return din, None


class _Add(Function):
    @staticmethod
    def forward(ctx, a, b):
        return ew(_dense_rows(a.detach()), _dense_rows(b.detach()), 2)

    @staticmethod
    def backward(ctx, g):
        return g, g


add = _Add.apply
layernorm = _LayerNorm.apply
dwconv = _DwConv.apply
spatial_gate16 = _SpatialGate16.apply
chan_mean = _ChanMean.apply
scale_channels = _ScaleChannels.apply
chanconv9 = _ChanConv9.apply
colconv9 = _ColConv9.apply
seq_attn = _SeqAttn.apply
flow_warp = _FlowWarp.apply
resample2 = _Resample2.apply
conv_last = _ConvLast.apply


def small_conv16(x, weight, bias, stride, pad, out_pad=0, transposed=False, act=ACT_NONE):
    return _SmallConv16.apply(x, weight, bias, stride, pad, out_pad, transposed, act)


def channel_attention(q, k, v, temp, heads):
    return _ChannelAttention.apply(q, k, v, temp, heads)


def mul_mask(x, mask, inv=False, anchor=None):
    return _MulMask.apply(x, mask, inv, anchor)
