"""Training-mode forward of ``CVSR_V8`` (arch/SIDECVSR_our.py:4406-4481) under torch autograd.

``CVSR_V8.forward`` dispatches here whenever gradients are enabled and a parameter requires them (train_LD_37.py:376-381:
``sr, _ = model(...)``; ``loss.backward()``).  The computation is the reference's, operator by operator; its convolutions run in
``cdfo_amd.autograd.CONV_PREC`` -- split-bf16 three-pass MFMA products by default (forward <= 3e-5 against the reference), exact
fp32 with ``CDFO_TRAIN_EXACT=1`` (the kernels of the ``precision="f32"`` inference mode: <= 1e-5) -- and everything else in exact
fp32 (which kernel families the switch reaches is listed next to ``CONV_PREC``), composed from the
``torch.autograd.Function`` objects of ``cdfo_amd/autograd.py`` -- HIP kernels in both directions, no CPU or ATen fallback
for anything pixel-sized.  The inference path's algebraic fusions (folded attention weights, composed stride-2 convolution,
fp16 tensors) are not used here: gradients are taken of the plain operator graph.

The hard Gumbel mask (arch.py:2191-2195) is built with ``masked_fill``: no gradient flows into it, so the mask generator
(``RDAB.conv_du_re`` / ``conv_du_re2``) receives zero gradients -- as it does in the reference -- and runs outside the graph.
``MV_deform_align.fusion_in`` is never called by the reference forward either (arch.py:3441-3444): its gradient stays None."""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F

from . import autograd as A
from . import kernels as K
from .kernels import ACT_LRELU, ACT_NONE, ACT_RELU

NF, NFRAMES = 64, 7


def _udsa(P, x2, res):
    u = "transformer_feature_extraction.path1.side_to_feaoneUDSA.body."
    t = A.conv(x2, P[u + "0.weight"], P[u + "0.bias"], 1, 1, ACT_LRELU)
    t = A.small_conv16(t, P[u + "2.weight"], P[u + "2.bias"], 2, 2, 0, False, ACT_LRELU)
    t = A.small_conv16(t, P[u + "4.weight"], P[u + "4.bias"], 2, 2, 0, False, ACT_LRELU)
    t = A.spatial_gate16(t, P[u + "6.spatial.weight"], P[u + "6.spatial.bias"])
    t = A.small_conv16(t, P[u + "7.weight"], P[u + "7.bias"], 2, 2, 0, True, ACT_LRELU)
    t = A.small_conv16(t, P[u + "9.weight"], P[u + "9.bias"], 2, 2, 1, True, ACT_LRELU)
    return A.conv(t, P[u + "11.weight"], P[u + "11.bias"], 1, 1, ACT_LRELU, res=[res])


def _feature_extraction(P, x1, x2):
    p = "transformer_feature_extraction.path1."
    for rnd in range(3):
        x2 = _udsa(P, x2, x1 if rnd == 0 else x2)
        ln = A.layernorm(x1, P[p + "norm1.body.weight"], P[p + "norm1.body.bias"])
        qkv = A.dwconv(A.conv(ln, P[p + "attn.qkv.weight"]), P[p + "attn.qkv_dwconv.weight"])
        att = A.channel_attention(qkv[..., 0:64], qkv[..., 64:128], qkv[..., 128:192], P[p + "attn.temperature"], 8)
        x1 = A.conv(att, P[p + "attn.project_out.weight"], res=[x1])
        ln = A.layernorm(x1, P[p + "norm2.body.weight"], P[p + "norm2.body.bias"])
        x1 = A.conv(ln, P[p + "conv.weight"], P[p + "conv.bias"], 1, 1, res=[x1, x2])
    return x1


def _rdab(P, res, x, noise, capture):
    r = "RDAB."
    B, H, W, _ = x.shape
    with torch.no_grad():          # the mask generator: no gradient reaches it through the hard threshold (module docstring)
        t = K.conv(res.detach(), K.pack_conv(P[r + "conv_du_re.0.weight"].detach(), P[r + "conv_du_re.0.bias"].detach()), act=ACT_RELU)
        t = K.conv(t, K.pack_conv(P[r + "conv_du_re.2.weight"].detach(), P[r + "conv_du_re.2.bias"].detach()), stride=2, pad=2, act=ACT_RELU)
        part, n = K.chan_sum_partial(t)
        vmax = K.vec_mlp(part, n, t.shape[1] * t.shape[2], P[r + "conv_du_re2.0.weight"].detach().contiguous(),
                         P[r + "conv_du_re2.0.bias"].detach().contiguous(), 64, ACT_RELU)
        mask = A.gumbel_mask(vmax, noise, B, H, W, capture)
    # zero-gradient anchor of the generator's parameters (parameter-sized arithmetic): their .grad becomes zeros, as in the reference
    anchor = sum((P[r + k].sum() * 0.0 for k in ("conv_du_re.0.weight", "conv_du_re.0.bias", "conv_du_re.2.weight", "conv_du_re.2.bias",
                                                 "conv_du_re2.0.weight", "conv_du_re2.0.bias")), torch.zeros((), device=x.device))
    xq = A.conv(x, P[r + "input_conv.weight"], P[r + "input_conv.bias"])
    q4, v4 = xq[..., 0:64], xq[..., 64:128]
    sq = A.chanconv9(A.mul_mask(q4, mask, False, anchor), P[r + "directW1_conv.weight"], P[r + "directW1_conv.bias"])
    vv = A.chanconv9(v4, P[r + "directW1_conv.weight"], P[r + "directW1_conv.bias"])
    rowo = A.seq_attn(sq, vv, 0)
    qc = A.colconv9(sq, P[r + "directH1_conv.weight"], P[r + "directH1_conv.bias"])
    long_out = A.seq_attn(qc, rowo, 1)
    loc = A.seq_attn(A.mul_mask(q4, mask, True), v4, 2)
    return A.conv([long_out, loc], P[r + "fuse.weight"], P[r + "fuse.bias"], res=[x])


def _gate(P, prefix, z, act0=ACT_RELU):
    """conv_du(avg_pool(z)): the 64-vector MLP per image is torch arithmetic on [B, 64] tensors."""
    m = A.chan_mean(z)
    w0, b0, w2, b2 = (P[prefix + k] for k in ("0.weight", "0.bias", "2.weight", "2.bias"))
    h = F.relu(F.linear(m, w0.flatten(1), b0))
    return torch.sigmoid(F.linear(h, w2.flatten(1), b2))


def _resblock(P, p, x, extra_res=()):
    t = A.conv(x, P[p + "conv1.weight"], P[p + "conv1.bias"], 1, 1, ACT_RELU)
    return A.conv(t, P[p + "conv2.weight"], P[p + "conv2.bias"], 1, 1, res=[x, *extra_res])


def _align(P, xc, extra, pred, mv, mv_bstride):
    a = "MV_deform_align."
    warped = A.flow_warp(extra, mv, mv_bstride)
    k = A.conv([warped, pred], P[a + "fusion_out.0.weight"], None, act=ACT_RELU)
    temp = P[a + "temperature"]
    v1 = A.scale_channels(warped, _gate(P, a + "conv_du.", warped))
    v2 = A.scale_channels(pred, _gate(P, a + "conv_du.", pred))
    o1 = A.conv(A.channel_attention(xc, k, v1, temp, 4), P[a + "project_out.weight"])
    o12 = A.conv(A.channel_attention(xc, k, v2, temp, 4), P[a + "project_out.weight"], res=[o1])
    out = A.conv([o12, xc], P[a + "fusion_out.0.weight"], None, act=ACT_RELU)
    out = A.scale_channels(out, _gate(P, a + "CALayer.conv_du.", out))
    out = _resblock(P, a + "ResidualBlock.", out)
    return _resblock(P, a + "ResidualBlock1.", out, extra_res=[xc])


def _block(P, p, x):
    def body(z, res=()):
        t = A.conv(z, P[p + "body.0.weight"], P[p + "body.0.bias"], 1, 1, ACT_LRELU)
        return A.conv(t, P[p + "body.2.weight"], P[p + "body.2.bias"], 1, 1, res=list(res))

    down = lambda z: A.resample2(A.conv(z, P[p + "down.0.weight"], P[p + "down.0.bias"]), False)   # noqa: E731
    up = lambda z: A.resample2(A.conv(z, P[p + "up.0.weight"], P[p + "up.0.bias"]), True)          # noqa: E731
    y = body(x, res=[x])
    y = A.add(y, up(body(down(x))))
    return A.add(y, down(body(up(x))))


def _trunk(P, x):
    y = x
    for g in range(7):
        gp = f"recon_trunk.body.{g}."
        r = y
        for b in range(3):
            r = _block(P, gp + f"body.{b}.", r)
        y = A.conv(r, P[gp + "conv.weight"], P[gp + "conv.bias"], 1, 1, res=[y] + ([x] if g == 6 else []))
    return y


def _pixel_shuffle_nhwc(t: torch.Tensor) -> torch.Tensor:
    """F.pixel_shuffle(., 2) in pixel-major layout: out[b, 2y+dy, 2x+dx, c] = in[b, y, x, c*4 + dy*2 + dx] (views + one copy)."""
    B, H, W, C4 = t.shape
    c = C4 // 4
    return t.view(B, H, W, c, 2, 2).permute(0, 1, 4, 2, 5, 3).reshape(B, 2 * H, 2 * W, c)


def forward_train(model, x, mvs0, mvs1, pms, rms, ufs, noise):
    """(out [B,1,4H,4W], L1_fea [B*7,64,H,W]) with the autograd graph attached.  `noise`: six [B,64,H,W] uniform tensors, or
    None to draw them inside the mask kernel (seeded per forward like the inference path)."""
    P: Dict[str, torch.Tensor] = dict(model.named_parameters())
    B, N, C, H, W = x.shape
    ctr = N // 2
    # Envelope: the backward kernels of this path are sized for the training scripts' crops (train_LD_37.py:39,316-325: 64 x 64).
    # Several are serial by construction at large frames -- the spatial gate's weight gradient runs 99 workgroups over all pixels,
    # the attention backward re-reads a sequence's keys for every query pair (O(L^2) global reads) -- so a grad-enabled forward at
    # validation size works but is slow; say so once instead of silently entering it (evaluation belongs under torch.no_grad()).
    if H * W > 256 * 256 and not getattr(model, "_warned_train_size", False):
        import warnings
        warnings.warn(f"CVSR_V8 (HIP): autograd forward at {H}x{W}: the training path is tuned for 64x64 crops; wrap evaluation in "
                      "torch.no_grad() to get the fused inference schedule")
        model._warned_train_size = True
    x = x.contiguous().float()
    pms = pms.contiguous().float()
    mvs1 = mvs1.contiguous().float()
    if ufs.shape[1] != 1:
        ufs, rms = ufs.transpose(1, 2), rms.transpose(1, 2)
    ufs, rms = ufs.contiguous().float(), rms.contiguous().float()
    Pn = H * W

    f = A.stem(x.view(B * N, H, W), P["conv_first.weight"], P["conv_first.bias"], ACT_LRELU)
    s = A.stem(pms.view(B * N, H, W), P["conv_second.weight"], P["conv_second.bias"])
    L1 = _feature_extraction(P, f, s)                                  # [B*7, H, W, 64], clip-major
    fea = L1.view(B, N, H, W, NF)

    aligned: List[torch.Tensor] = []
    draw = 0
    for i in range(N):
        if i == ctr:
            aligned.append(fea[:, ctr])
            continue
        ufs_prior = A.stem(ufs[:, 0, i], P["conv_expand_ufs.weight"], P["conv_expand_ufs.bias"])
        rms_prior = A.stem(rms[:, 0, i], P["conv_expand_rms.weight"], P["conv_expand_rms.bias"])
        fea_com = A.add(fea[:, i], rms_prior)
        nz = ("rng", model._noise_seed, draw) if noise is None else noise[draw].to(x.device).float().contiguous()
        x_n = _rdab(P, rms_prior, fea_com, nz, model.capture_noise)
        fea_i = A.conv([fea[:, i], x_n], P["conv_expand_fea_r.weight"], P["conv_expand_fea_r.bias"], 1, 1)
        aligned.append(_align(P, fea[:, ctr], fea_i, ufs_prior, mvs1[:, i], N * 2 * Pn))
        draw += 1

    fused = A.conv(aligned, P["tsa_fusion.weight"], P["tsa_fusion.bias"], act=ACT_LRELU)
    t = _trunk(P, fused)
    t = _pixel_shuffle_nhwc(A.conv(t, P["upconv1.weight"], P["upconv1.bias"], act=ACT_LRELU))
    t = _pixel_shuffle_nhwc(A.conv(t, P["upconv2.weight"], P["upconv2.bias"], act=ACT_LRELU))
    out = A.conv_last(t, P["conv_last.weight"], P["conv_last.bias"], x[:, ctr], N * Pn)
    return out, L1.permute(0, 3, 1, 2)
