"""Training-mode forward of ``CVSR_V7`` (arch/SIDECVSR_our.py:4215-4367) under torch autograd.

``CVSR_V7.forward`` dispatches here whenever gradients are enabled and a parameter requires them -- the reference class is an
ordinary trainable ``nn.Module``.  Same construction as ``cvsr_v8_train.py``: the reference's operator graph, operator by operator,
from ``torch.autograd.Function``s whose forward AND backward run in libcdfo_hip.so -- the pixel-major Functions of
``cdfo_amd/autograd.py`` (convolutions, LayerNorm, depthwise, channel attention, resampling, stems), the DCN-aligned
``MVDualAttAlignment.forward_train`` (``cdfo_dcn_backward``), and the five V7-only pieces below (csrc/v7_train.hip): channel pooling,
the spatial gates, the SOFT Gumbel softmax of ``RDAB`` (arch.py:2813-2847 -- unlike V8's hard mask it carries a gradient into the
mask generator, whose stride-2 convolution therefore needs an input gradient here) and their adjoints.  torch supplies the graph,
views / copies, and arithmetic on parameter-sized or per-image tensors ([B, 64] gate vectors, weight paddings); nothing of size
O(pixels) is computed by ATen.  Convolution arithmetic = ``cdfo_amd.autograd.CONV_PREC``.  Tuned for training crops, not for
validation-size frames (the 2 -> 1 channel gate convolutions run on the one-thread-per-output NCHW kernels)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List

import torch
import torch.nn.functional as F
from torch.autograd import Function

from . import _lib
from . import autograd as A
from . import kernels as K
from . import nchw_autograd as G
from ._lib import check
from .cvsr_v8_train import _pixel_shuffle_nhwc
from .kernels import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, _stream, _vp

NF, NFRAMES = 64, 7


def _rows(t):
    return A._dense_rows(t.detach())


# ------------------------------------------------------------------------------------------------ V7-only Functions
class _ChanPool(Function):
    """ChannelPool (arch.py:1883-1885): [B,H,W,64] -> [B,H,W,2] = (max_c, mean_c)."""

    @staticmethod
    def forward(ctx, x):
        x = _rows(x)
        ctx.save_for_backward(x)
        return K.chan_pool(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = g.contiguous()
        dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        npix = x.shape[0] * x.shape[1] * x.shape[2]
        check(_lib.lib().cdfo_chan_pool_bwd(_vp(x), x.stride(-2), _vp(g), C.c_longlong(npix), _vp(dx), 64, _stream()), "cdfo_chan_pool_bwd")
        return dx


def _mul_plane(x, plane):
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    npix = x.shape[0] * x.shape[1] * x.shape[2]
    check(_lib.lib().cdfo_mul_plane(_vp(x), x.stride(-2), _vp(plane), C.c_longlong(npix), _vp(out), 64, _stream()), "cdfo_mul_plane")
    return out


class _MulPlane(Function):
    """x [B,H,W,64] * plane [B,H,W] (a spatial gate broadcast over the channels, arch.py:2729)."""

    @staticmethod
    def forward(ctx, x, plane):
        x, plane = _rows(x), plane.detach().contiguous()
        ctx.save_for_backward(x, plane)
        return _mul_plane(x, plane)

    @staticmethod
    def backward(ctx, g):
        x, plane = ctx.saved_tensors
        g = A._c(g)
        dplane = torch.empty(plane.shape, dtype=torch.float32, device=x.device)
        npix = plane.numel()
        check(_lib.lib().cdfo_dot_plane(_vp(g), 64, _vp(x), x.stride(-2), C.c_longlong(npix), _vp(dplane), _stream()), "cdfo_dot_plane")
        return _mul_plane(g, plane), dplane


class _Mul(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _rows(a), _rows(b)
        ctx.save_for_backward(a, b)
        return A.ew(a, b, 0)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = A._c(g)
        return A.ew(g, b, 0), A.ew(g, a, 0)


class _GumbelSoftmax(Function):
    """softmax_c(v[b][c] + Gumbel(u)) per pixel (RDAB.gumbel_softmax, arch.py:2813-2822): v [B,64] (the broadcast v_max map), u the
    uniform draw [B,64,H,W] (no gradient) -> [B,H,W,64]."""

    @staticmethod
    def forward(ctx, v, u):
        B, _, H, W = u.shape
        out = torch.empty((B, H, W, 64), dtype=torch.float32, device=u.device)
        check(_lib.lib().cdfo_gumbel_softmax(_vp(v.detach().contiguous()), _vp(u.contiguous()), B, C.c_longlong(H * W), _vp(out), 64,
                                             _stream()), "cdfo_gumbel_softmax")
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (r,) = ctx.saved_tensors
        g = A._c(g)
        dz = torch.empty_like(r)
        npix = r.shape[0] * r.shape[1] * r.shape[2]
        check(_lib.lib().cdfo_softmax64_bwd(_vp(r), 64, _vp(g), 64, C.c_longlong(npix), _vp(dz), 64, _stream()), "cdfo_softmax64_bwd")
        return A.coldot(dz, None, r.shape[0]), None


class _ConvS2(Function):
    """relu(conv3x3(x; stride 2, pad 2)) with an INPUT gradient (RDAB.conv_du_re.2, arch.py:2806-2809; V8's copy of this layer sits
    behind a hard threshold and needs none).  The input gradient of a strided convolution is a stride-1 correlation of the
    zero-stuffed output gradient with the flipped, transposed weights: dx[i] = sum_k' g^[i + k'] w[2 - k'], g^[2 o] = g[o]."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = _rows(x)
        y = K.conv([x], K.pack_conv(weight.detach(), bias.detach()), stride=2, pad=2, act=ACT_RELU, prec=K.PREC_F32)
        ctx.save_for_backward(x, weight, y)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, y = ctx.saved_tensors
        gp = A.act_bwd(A._c(g), y, ACT_RELU)
        B, H, W, Cc = x.shape
        _, Ho, Wo, Co = gp.shape
        dW = torch.empty_like(weight)
        A.conv_wgrad(gp, x, 3, 2, 2, dW, Cc, 0)
        db = A.coldot(gp, None, 1).view(-1)
        stuffed = gp.new_zeros((B, H + 2, W + 2, Co))                 # a copy, no arithmetic: g^ with its zero frame
        stuffed[:, 0:2 * Ho:2, 0:2 * Wo:2] = gp
        wt = weight.detach().flip(2, 3).transpose(0, 1).contiguous()
        dx = K.conv([stuffed], K.pack_conv(wt, None), stride=1, pad=0, prec=K.PREC_F32)
        return dx, dW, db


chan_pool = _ChanPool.apply
mul_plane = _MulPlane.apply
mul = _Mul.apply
gumbel_softmax = _GumbelSoftmax.apply
conv_s2 = _ConvS2.apply


def _spatial_gate_plane(P, prefix, x, pad):
    """sigmoid(conv_kxk([max_c x, mean_c x])) as a [B,H,W] plane (SpatialAttention, arch.py:2719-2730; ChannelPool :1883-1885)."""
    pooled = G.to_nchw(chan_pool(x))                                   # [B,2,H,W]
    att = G.conv2d(pooled, P[prefix + "spatial.weight"], P[prefix + "spatial.bias"], 1, pad, ACT_SIGMOID)      # [B,1,H,W]
    return att.reshape(att.shape[0], att.shape[2], att.shape[3])


# ------------------------------------------------------------------------------------------------ the forward
def _feature_extraction(P, x1, x2):
    """PartitionTransformerBlock (arch.py:1340-1368): four weight-shared rounds."""
    p = "transformer_feature_extraction.path1."
    for _ in range(4):
        x2 = mul_plane(x2, _spatial_gate_plane(P, p + "SA.", x2, 3))
        ln = A.layernorm(x1, P[p + "norm1.body.weight"], P[p + "norm1.body.bias"])
        qkv = A.dwconv(A.conv(ln, P[p + "attn.qkv.weight"]), P[p + "attn.qkv_dwconv.weight"])
        att = A.channel_attention(qkv[..., 0:64], qkv[..., 64:128], qkv[..., 128:192], P[p + "attn.temperature"], 8)
        x1 = A.conv(att, P[p + "attn.project_out.weight"], res=[x1, x2])
        ln = A.layernorm(x1, P[p + "norm2.body.weight"], P[p + "norm2.body.bias"])
        x1 = A.conv(ln, P[p + "conv.weight"], P[p + "conv.bias"], 1, 1, res=[x1])
    return x1


def _rdab(P, res, xc, u):
    """RDAB.forward (arch.py:2830-2847) with the uniform draw u [B,64,h,w]."""
    r = "RDAB."
    t = A.conv(res, P[r + "conv_du_re.0.weight"], P[r + "conv_du_re.0.bias"], act=ACT_RELU)
    t = conv_s2(t, P[r + "conv_du_re.2.weight"], P[r + "conv_du_re.2.bias"])
    v = F.relu(F.linear(A.chan_mean(t), P[r + "conv_du_re2.0.weight"].flatten(1), P[r + "conv_du_re2.0.bias"]))     # [B,64]
    r_m = gumbel_softmax(v, u)
    att = _spatial_gate_plane(P, r, xc, 1)
    xf = A.conv(A.conv(xc, P[r + "conv_dc.0.weight"], P[r + "conv_dc.0.bias"], act=ACT_LRELU), P[r + "conv_dc.2.weight"], P[r + "conv_dc.2.bias"])
    mixed = A.add(mul(xf, r_m), mul_plane(xf, att))                      # x_f * (r_m + att)
    return A.conv(mixed, P[r + "conv_df.0.weight"], P[r + "conv_df.0.bias"], act=ACT_LRELU)


def _block(P, p, xs):
    """Block.forward over the level list (arch.py:367-375)."""
    def body(z):
        t = A.conv(z, P[p + "body.0.weight"], P[p + "body.0.bias"], 1, 1, ACT_LRELU)
        return A.conv(t, P[p + "body.2.weight"], P[p + "body.2.bias"], 1, 1)

    res = [body(z) for z in xs]
    down = [res[0]] + [A.resample2(A.conv(z, P[p + "down.0.weight"], P[p + "down.0.bias"]), False) for z in res[:-1]]
    up = [A.resample2(A.conv(z, P[p + "up.0.weight"], P[p + "up.0.bias"]), True) for z in res[1:]] + [res[-1]]
    return [A.add(A.add(x, r), A.add(d, u)) for x, r, d, u in zip(xs, res, down, up)]


def _trunk(P, xs):
    ys = xs
    for g in range(7):
        gp = f"recon_trunk.body.{g}."
        rs = ys
        for b in range(3):
            rs = _block(P, gp + f"body.{b}.", rs)
        ys = [A.conv(r, P[gp + "conv.weight"], P[gp + "conv.bias"], 1, 1, res=[y] + ([x] if g == 6 else []))
              for x, y, r in zip(xs, ys, rs)]
    return ys


def forward_train(model, x, mvs0, mvs1, pms, rms, ufs, noise):
    """(out [B,1,4H,4W], L1_fea [B*7,64,H,W]) with the autograd graph attached.  `noise`: the 36 uniform draws in the reference's
    call order (each [B,64,H>>lv,W>>lv]) or None to draw them with torch.rand like the inference path."""
    P: Dict[str, torch.Tensor] = dict(model.named_parameters())
    B, N, Cc, H, W = x.shape
    ctr = N // 2
    dev = x.device
    x = x.contiguous().float()
    pms = pms.contiguous().float()
    mvs = (mvs0.contiguous().float(), mvs1.contiguous().float())
    if ufs.shape[1] != 1:
        ufs, rms = ufs.transpose(1, 2), rms.transpose(1, 2)
    ufs, rms = ufs.contiguous().float(), rms.contiguous().float()
    Pn = H * W
    align = model.MV_deform_align

    f = A.stem(x.view(B * N, H, W), P["conv_first.weight"], P["conv_first.bias"], ACT_LRELU)
    s = A.stem(pms.view(B * N, H, W), P["conv_second.weight"], P["conv_second.bias"])
    L1 = _feature_extraction(P, f, s)
    pyr = [L1, A.resample2(L1, False)]
    pyr.append(A.resample2(pyr[1], False))

    draw = 0
    prev = None
    fused_pyr: List[torch.Tensor] = []
    for lv in (2, 1, 0):
        h, wd = H >> lv, W >> lv
        fea = pyr[lv].view(B, N, h, wd, NF)
        centre = fea[:, ctr]
        centre_nchw = G.to_nchw(centre)

        def neighbour(i, mv_all, u):
            if lv == 0:
                mv, u_img, r_img = mv_all[:, i].contiguous(), ufs[:, 0, i], rms[:, 0, i]
            else:
                mv = K.shrink_planes(mv_all[:, i], lv)
                u_img, r_img = K.shrink_planes(ufs[:, :, i], lv)[:, 0], K.shrink_planes(rms[:, :, i], lv)[:, 0]
            ufs_prior = A.stem(u_img.contiguous(), P["conv_expand_ufs.weight"], P["conv_expand_ufs.bias"])
            rms_prior = A.stem(r_img.contiguous(), P["conv_expand_rms.weight"], P["conv_expand_rms.bias"])
            fea_com = A.add(fea[:, i], rms_prior)
            if prev is not None:
                fea_com = A.add(fea_com, A.resample2(prev[i], True))
            x_n = _rdab(P, rms_prior, fea_com, u)
            fea_i = A.conv([fea[:, i], x_n], P["conv_expand_fea_r.weight"], P["conv_expand_fea_r.bias"], 1, 1)
            out = align.forward_train(centre_nchw, G.to_nchw(fea_i), G.to_nchw(ufs_prior), mv)
            return G.to_pixel_major(out)

        jobs = [(0, i) for i in range(N - 1, -1, -1) if i != ctr] + [(1, i) for i in range(N) if i != ctr]
        res = {}
        for n, (which, i) in enumerate(jobs):
            if noise is None:
                u = torch.rand((B, NF, h, wd), device=dev, dtype=torch.float32).clamp_min_(1e-30)
            else:
                u = noise[draw + n].to(device=dev, dtype=torch.float32).contiguous()
            res[(which, i)] = neighbour(i, mvs[which], u)
        draw += len(jobs)
        cur = [centre if i == ctr else A.conv([res[(0, i)], res[(1, i)]], P["fb_fusion.weight"], P["fb_fusion.bias"])
               for i in range(N)]
        prev = cur
        fused_pyr.append(A.conv(cur, P["tsa_fusion.weight"], P["tsa_fusion.bias"], act=ACT_LRELU))

    outs = _trunk(P, fused_pyr[::-1])
    o3 = A.conv(outs[2], P["upconv1_L3.weight"], P["upconv1_L3.bias"], act=ACT_LRELU)
    o3 = _pixel_shuffle_nhwc(_pixel_shuffle_nhwc(o3))                                              # [B,H,W,4]
    o2 = _pixel_shuffle_nhwc(A.conv(outs[1], P["upconv1_L2.weight"], P["upconv1_L2.bias"], act=ACT_LRELU))     # [B,H,W,16]
    # the 84-channel concatenation [64 | 16 | 4] as three 16-aligned sources: zero channels / zero weight columns (copies, no arithmetic)
    o3p = F.pad(o3, (0, 12))
    wu = torch.cat([P["upconv1.weight"], P["upconv1.weight"].new_zeros(256, 12, 1, 1)], 1)
    t = _pixel_shuffle_nhwc(A.conv([outs[0], o2, o3p], wu, P["upconv1.bias"], act=ACT_LRELU))
    t = _pixel_shuffle_nhwc(A.conv(t, P["upconv2.weight"], P["upconv2.bias"], act=ACT_LRELU))
    w3 = F.pad(P["conv_last.weight"], (1, 1, 1, 1))                                                # the 1x1 conv_last as the centre tap of a 3x3
    out = A.conv_last(t, w3, P["conv_last.bias"], x[:, ctr], N * Pn)
    return out, L1.permute(0, 3, 1, 2)
