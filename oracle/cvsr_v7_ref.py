"""ORACLE (test infrastructure, not product code): torch-cpu restatement of the reference's ``CVSR_V7.forward``
(arch/SIDECVSR_our.py:4215-4367), the DCN-aligned three-level-pyramid variant, over a plain
``state_dict``.  Only tests/ and oracle/gen_fixtures.py import this.

Follows, in the reference repo (arch/SIDECVSR_our.py unless noted):
  :4251-4272   stems, ``PAItransformer_feat_extract`` (:1602-1612) = ``PartitionTransformerBlock`` (:1340-1368): four
               weight-shared rounds of  x2 = SpatialAttention(x2) (:2719-2730, ChannelPool :1883-1885);
               x1 += MDTA(LN1(x1)) + x2;  x1 += conv3x3(LN2(x1))
  :4267-4272   feature pyramid: Interpolate(0.5) twice (bilinear, align_corners=False == 2x2 mean)
  :4275-4347   per level (coarse to fine), a backward pass (i = 6..0, MV = mvs0) and a forward pass (i = 0..6, MV =
               mvs1) over the six neighbours: priors resized by F.interpolate(scale 0.5 / 0.25) and divided by 2 / 4,
               ``RDAB`` (:2795-2847, soft Gumbel softmax over channels + 3x3 spatial gate), ``conv_expand_fea_r``,
               ``MVDualAttAlignment`` (:3303-3352 -> oracle/dcn_modules_ref.py), ``fb_fusion``; the previous level's
               fused features are added after Interpolate(2.0); ``tsa_fusion`` + LeakyReLU per level
  :4352-4366   ``SCNet`` over the three-level list (:337-467: Block / SCGroup / SCNet with cross-scale exchange),
               pyramid merge by 1x1 conv + pixel shuffle, the 84-channel ``upconv1``, ``upconv2``, 1x1 ``conv_last``,
               bilinear x4 skip
Pinned by golden vectors produced by the REAL reference class (oracle/gen_fixtures.py, tests/golden/cvsr_v7_*.npz)
whose only substitution is the un-vendored ``torchvision.ops.deform_conv2d`` -> oracle/dcn_ref.c."""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor

from .cvsr_v8_ref import NF, NFRAMES, _conv, _layernorm_c, _lrelu, _mdta, _half, _double, make_inputs
from .dcn_modules_ref import mv_dual_att_alignment_forward

N_DRAWS = 36     # RDAB calls per forward: 3 levels x (6 backward-pass + 6 forward-pass) neighbours


def spatial_attention(sd, p: str, x: Tensor, pad: int) -> Tensor:
    pooled = torch.cat([x.max(1, keepdim=True)[0], x.mean(1, keepdim=True)], 1)
    return torch.sigmoid(_conv(sd, p + "spatial", pooled, 1, pad))


def feature_extraction_v7(sd, x1: Tensor, x2: Tensor) -> Tensor:
    p = "transformer_feature_extraction.path1."
    for _ in range(4):
        x2 = x2 * spatial_attention(sd, p + "SA.", x2, 3)
        x1 = x1 + _mdta(sd, p + "attn.", _layernorm_c(sd, p + "norm1", x1)) + x2
        x1 = x1 + _conv(sd, p + "conv", _layernorm_c(sd, p + "norm2", x1), 1, 1)
    return x1


def rdab_v7(sd, res: Tensor, x_c: Tensor, u: Tensor) -> Tensor:
    """``RDAB.forward`` (arch.py:2830-2847) with the uniform draw ``u`` supplied by the caller."""
    p = "RDAB."
    r_f = F.relu(_conv(sd, p + "conv_du_re.2", F.relu(_conv(sd, p + "conv_du_re.0", res)), 2, 2))
    v = F.relu(_conv(sd, p + "conv_du_re2.0", r_f.mean((2, 3), keepdim=True)))
    v = v.expand(-1, -1, res.shape[2], res.shape[3])          # bilinear resize of a 1x1 map == broadcast
    g = -(-u.log()).log()
    r_m = (v + g).softmax(1)
    att = spatial_attention(sd, p, x_c, 1)
    x_f = _conv(sd, p + "conv_dc.2", _lrelu(_conv(sd, p + "conv_dc.0", x_c)))
    return _lrelu(_conv(sd, p + "conv_df.0", x_f * (r_m + att)))


def _sub(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def scnet_pyramid(sd, xs: List[Tensor], groups: int = 7, blocks: int = 3) -> List[Tensor]:
    def block(p, xl):
        res = [_conv(sd, p + "body.2", _lrelu(_conv(sd, p + "body.0", z, 1, 1)), 1, 1) for z in xl]
        down = [res[0]] + [_half(_conv(sd, p + "down.0", z)) for z in res[:-1]]
        up = [_double(_conv(sd, p + "up.0", z)) for z in res[1:]] + [res[-1]]
        return [x + r + d + u for x, r, d, u in zip(xl, res, down, up)]

    ys = xs
    for g in range(groups):
        p = f"recon_trunk.body.{g}."
        rs = ys
        for b in range(blocks):
            rs = block(p + f"body.{b}.", rs)
        ys = [y + _conv(sd, p + "conv", r, 1, 1) for y, r in zip(ys, rs)]
    return [x + y for x, y in zip(xs, ys)]


def cvsr_v7_forward(sd: Dict[str, Tensor], x: Tensor, mvs0: Tensor, mvs1: Tensor, pms: Tensor, rms: Tensor, ufs: Tensor,
                    pre_L1_fea: Optional[Tensor] = None, gumbel_u: Optional[List[Tensor]] = None,
                    taps: Optional[dict] = None):
    """Returns ``(out [B,1,4H,4W], L1_fea [B*7,64,H,W])``.  ``gumbel_u``: the 36 uniform draws in call order (level 2
    backward pass i = 6,5,4,2,1,0, level 2 forward pass i = 0,1,2,4,5,6, then level 1, then level 0), each
    [B,64,H>>lv,W>>lv]; None draws them like the reference."""
    B, N, C, H, W = x.shape
    ctr = N // 2
    if pre_L1_fea is None:
        f = _lrelu(_conv(sd, "conv_first", x.reshape(-1, C, H, W), 1, 1))
        s = _conv(sd, "conv_second", pms.reshape(-1, C, H, W), 1, 1)
        L1 = feature_extraction_v7(sd, f, s)
    else:
        f = _lrelu(_conv(sd, "conv_first", x[:, -1], 1, 1))
        s = _conv(sd, "conv_second", pms[:, -1], 1, 1)
        new = feature_extraction_v7(sd, f, s).unsqueeze(1)
        L1 = torch.cat([pre_L1_fea.view(B, N, -1, H, W)[:, 1:], new], 1).reshape(B * N, -1, H, W)
    pyr = [L1, _half(L1)]
    pyr.append(_half(pyr[1]))
    if ufs.shape[1] != 1:
        ufs, rms = ufs.transpose(1, 2), rms.transpose(1, 2)
    align_sd = _sub(sd, "MV_deform_align.")
    draw = 0
    prev = None                                   # previous (coarser) level's fused neighbours [B,N,64,h,w]
    fused_pyr = []
    for lv in (2, 1, 0):
        h, w = H >> lv, W >> lv
        fea = pyr[lv].view(B, N, -1, h, w)

        def shrink(t):
            if lv == 0:
                return t
            return F.interpolate(t, scale_factor=0.5 ** lv, mode="bilinear", align_corners=False) / float(2 ** lv)

        def neighbour(i, mvs):
            nonlocal draw
            mv = shrink(mvs[:, i])
            ufs_prior = _conv(sd, "conv_expand_ufs", shrink(ufs[:, :, i]), 1, 1)
            rms_prior = _conv(sd, "conv_expand_rms", shrink(rms[:, :, i]), 1, 1)
            fea_com = fea[:, i] + rms_prior
            if prev is not None:
                fea_com = fea_com + _double(prev[:, i])
            if gumbel_u is None:
                u = torch.rand_like(rms_prior)
                while bool((u == 0).any()):
                    u = torch.rand_like(rms_prior)
            else:
                u = gumbel_u[draw]
            draw += 1
            x_n = rdab_v7(sd, rms_prior, fea_com, u)
            fea_i = _conv(sd, "conv_expand_fea_r", torch.cat([fea[:, i], x_n], 1), 1, 1)
            return mv_dual_att_alignment_forward(align_sd, fea[:, ctr], fea_i, ufs_prior, mv, 10.0, 16)

        back = {i: neighbour(i, mvs0) for i in range(N - 1, -1, -1) if i != ctr}
        cur = []
        for i in range(N):
            if i == ctr:
                cur.append(fea[:, i])
            else:
                cur.append(_conv(sd, "fb_fusion", torch.cat([back[i], neighbour(i, mvs1)], 1)))
        prev = torch.stack(cur, 1)
        fused = _lrelu(_conv(sd, "tsa_fusion", prev.reshape(B, -1, h, w)))
        if taps is not None:
            taps[f"fused_L{lv + 1}"] = fused
        fused_pyr.append(fused)
    outs = scnet_pyramid(sd, fused_pyr[::-1])
    if taps is not None:
        taps["L1_fea"] = L1
        taps["trunk_L1"] = outs[0]
    o3 = F.pixel_shuffle(F.pixel_shuffle(_lrelu(_conv(sd, "upconv1_L3", outs[2])), 2), 2)
    o2 = F.pixel_shuffle(_lrelu(_conv(sd, "upconv1_L2", outs[1])), 2)
    out = _lrelu(F.pixel_shuffle(_conv(sd, "upconv1", torch.cat([outs[0], o2, o3], 1)), 2))
    out = _lrelu(F.pixel_shuffle(_conv(sd, "upconv2", out), 2))
    out = _conv(sd, "conv_last", out)
    out = out + F.interpolate(x[:, ctr], scale_factor=4.0, mode="bilinear", align_corners=False)
    return out, L1


# --------------------------------------------------------------------------- deterministic weights / inputs
def state_dict_spec_v7() -> List[tuple]:
    """(key, shape, fan_in-or-None, mode) for the 247 entries of the reference ``CVSR_V7().state_dict()``."""
    spec: List[tuple] = []

    def conv(key, co, ci, k, bias=True, mode="default"):
        spec.append((key + ".weight", (co, ci, k, k), ci * k * k, mode))
        if bias:
            spec.append((key + ".bias", (co,), ci * k * k, "bias0" if mode == "kaiming0.1" else "bias"))

    for k in ("conv_first", "conv_second"):
        conv(k, 64, 1, 3)
    p = "transformer_feature_extraction.path1."
    spec.append((p + "norm1.body.weight", (64,), None, "ones"))
    spec.append((p + "norm1.body.bias", (64,), None, "zeros"))
    spec.append((p + "attn.temperature", (8, 1, 1), None, "ones"))
    conv(p + "attn.qkv", 192, 64, 1, bias=False)
    conv(p + "attn.qkv_dwconv", 192, 1, 3, bias=False)
    conv(p + "attn.project_out", 64, 64, 1, bias=False)
    spec.append((p + "norm2.body.weight", (64,), None, "ones"))
    spec.append((p + "norm2.body.bias", (64,), None, "zeros"))
    conv(p + "conv", 64, 64, 3)
    conv(p + "SA.spatial", 1, 2, 7)
    conv("conv_expand_fea_r", 64, 128, 3)
    conv("conv_expand_ufs", 64, 1, 3)
    conv("conv_expand_rms", 64, 1, 3)
    conv("fb_fusion", 64, 128, 1)
    conv("tsa_fusion", 64, 448, 1)
    for g in range(7):
        gp = f"recon_trunk.body.{g}."
        conv(gp + "conv", 64, 64, 3)
        for b in range(3):
            bp = gp + f"body.{b}."
            conv(bp + "body.0", 256, 64, 3, mode="kaiming0.1")
            conv(bp + "body.2", 64, 256, 3, mode="kaiming0.1")
            conv(bp + "down.0", 64, 64, 1, mode="kaiming0.1")
            conv(bp + "up.0", 64, 64, 1, mode="kaiming0.1")
    conv("upconv1", 256, 84, 1)
    conv("upconv2", 256, 64, 1)
    conv("conv_last", 1, 64, 1)
    a = "MV_deform_align."
    conv(a[:-1], 64, 64, 3)                                   # the DCN's own weight / bias
    spec.append((a + "temperature", (8, 1, 1), None, "ones"))
    conv(a + "conv_offset_mask", 432, 64, 3)                  # parameters of the parent class, unused by forward
    conv(a + "conv_offset.0", 64, 64, 3)
    conv(a + "conv_offset.2", 432, 64, 3, mode="offset_head")
    conv(a + "conv_du.0", 4, 64, 1)
    conv(a + "conv_du.2", 64, 4, 1)
    conv(a + "fusion_out", 64, 128, 1, bias=False)
    conv(a + "project_out", 64, 64, 1, bias=False)
    r = "RDAB."
    conv(r + "conv_du_re.0", 64, 64, 1)
    conv(r + "conv_du_re.2", 64, 64, 3)
    conv(r + "conv_du_re2.0", 64, 64, 1)
    conv(r + "conv_dc.0", 64, 64, 1)
    conv(r + "conv_dc.2", 64, 64, 1)
    conv(r + "spatial", 1, 2, 3)
    conv(r + "conv_df.0", 64, 64, 1)
    conv("upconv1_L2", 64, 64, 1)
    conv("upconv1_L3", 64, 64, 1)
    return spec


def make_state_dict_v7(seed: int = 0) -> Dict[str, Tensor]:
    """Deterministic weights (numpy RandomState).  The offset head ``conv_offset.2`` is zero-initialised in the
    reference (arch.py:3299-3300); here it gets a small non-zero init so the deformable sampling is exercised."""
    rs = np.random.RandomState(seed)
    sd: Dict[str, Tensor] = {}
    for key, shape, fan_in, mode in state_dict_spec_v7():
        if mode in ("default", "bias"):
            bound = 1.0 / math.sqrt(fan_in)
            a = rs.uniform(-bound, bound, size=shape)
        elif mode == "offset_head":
            a = rs.uniform(-1.0, 1.0, size=shape) * (0.3 / math.sqrt(fan_in))
        elif mode == "kaiming0.1":
            a = rs.standard_normal(size=shape) * (0.1 * math.sqrt(2.0 / fan_in))
        elif mode == "bias0":
            a = rs.uniform(-0.02, 0.02, size=shape)
        elif mode == "ones":
            a = 1.0 + rs.uniform(-0.2, 0.2, size=shape)
        elif mode == "zeros":
            a = rs.uniform(-0.1, 0.1, size=shape)
        else:
            raise ValueError(mode)
        sd[key] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return sd


def make_inputs_v7(B: int, H: int, W: int, seed: int, layout: str = "b1n"):
    """The V8 synthetic clip plus what only V7 reads: a non-zero ``mvs0`` (the backward-pass motion
    field: an independent block-constant field with the opposite temporal scaling) and the 36 Gumbel uniforms."""
    d = make_inputs(B, H, W, seed, layout)
    rs = np.random.RandomState(seed + 7919)
    hb, wb = (H + 7) // 8, (W + 7) // 8
    m = rs.randint(-64, 64, size=(B, 2, hb, wb)).astype(np.float32)
    base = np.repeat(np.repeat(m, 8, axis=2), 8, axis=3)[:, :, :H, :W] / 128.0
    scale = np.array([-3, -2, -1, 0, 1, 2, 3], dtype=np.float32).reshape(1, NFRAMES, 1, 1, 1)
    d["mvs0"] = torch.from_numpy(np.ascontiguousarray(base[:, None] * scale))
    gum = []
    for lv in (2, 1, 0):
        for _ in range(12):
            u = rs.random_sample((B, NF, H >> lv, W >> lv)).astype(np.float32)
            u[u == 0] = 0.5
            # float64 draws just below 1 round to 1.0f (about 3e-8 of them: a dozen per clip at 272x480): -log(-log 1) = +inf and the
            # soft Gumbel softmax of RDAB turns into NaN -- torch.rand_like, what the reference draws with, never returns 1
            u[u >= 1.0] = 0.5
            gum.append(torch.from_numpy(u))
    d["gumbel_u"] = gum
    return d
