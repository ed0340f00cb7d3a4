"""ORACLE (test infrastructure, not product code).

CPU restatement, in plain functional PyTorch over a ``state_dict``, of the
reference's ``CVSR_V8`` seven-frame x4 VSR forward pass.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this file; the product path (``cdfo_amd``) never does.

Pinned: ``oracle/gen_fixtures.py`` runs the real reference
(``/root/reference/arch/SIDECVSR_our.py``, imported read-only in the build
container) on seeded inputs/weights/noise and commits its outputs under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this restatement
against them.

Reference lines followed (``arch.py`` = ``arch/SIDECVSR_our.py``):
  forward orchestration ........ arch.py:4406-4481
  stems / fusion / upsampler ... arch.py:4379-4393
  PartitionTransformerSA_2 ..... arch.py:1441-1475
  WithBias LayerNorm ........... arch.py:1169-1198, 1218-1223
  MDTA ``Attention`` ........... arch.py:1545-1576
  side_to_feaoneUDSA_2 ......... arch.py:1815-1875 (+ SpatialAttention 2719-2730)
  LLongRangAttention ........... arch.py:2141-2249 (gumbel 2168-2177)
  flow_warp .................... arch.py:3068-3099
  DualAttAlignment ............. arch.py:3427-3496 (CALayer 2027-2043, ResidualBlock_noBN 254-271)
  SCNet_/SCGroup_/Block_ ....... arch.py:468-480, 430-444, 378-406 (Interpolate 324-333)

The one deliberate extension: the six uniform noise tensors that
``gumbel_softmax`` draws with ``torch.rand_like`` (arch.py:2169) can be passed
in (``gumbel_u``) so that both sides of a parity test see the same noise.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
NF = 64
NFRAMES = 7


# --------------------------------------------------------------------------- helpers
def _conv(sd: Dict[str, Tensor], key: str, x: Tensor, stride: int = 1, pad: int = 0) -> Tensor:
    return F.conv2d(x, sd[key + ".weight"], sd.get(key + ".bias"), stride=stride, padding=pad)


def _convT(sd, key, x, stride, pad, out_pad=0):
    return F.conv_transpose2d(x, sd[key + ".weight"], sd.get(key + ".bias"), stride=stride,
                              padding=pad, output_padding=out_pad)


def _lrelu(x: Tensor) -> Tensor:
    return F.leaky_relu(x, 0.1)


def _layernorm_c(sd, key, x: Tensor) -> Tensor:
    """Per-pixel LayerNorm over channels, biased variance, eps 1e-5 (arch.py:1169-1185)."""
    mu = x.mean(1, keepdim=True)
    var = x.var(1, keepdim=True, unbiased=False)
    w = sd[key + ".body.weight"].view(1, -1, 1, 1)
    b = sd[key + ".body.bias"].view(1, -1, 1, 1)
    return (x - mu) / torch.sqrt(var + 1e-5) * w + b


def _channel_attention(q: Tensor, k: Tensor, v: Tensor, heads: int, temperature: Tensor) -> Tensor:
    """Transposed (channel x channel) attention over all pixels (arch.py:1555-1571, 3463-3472)."""
    b, c, h, w = q.shape
    q = q.reshape(b, heads, c // heads, h * w)
    k = k.reshape(b, heads, c // heads, h * w)
    v = v.reshape(b, heads, c // heads, h * w)
    q = F.normalize(q, dim=-1)
    k = F.normalize(k, dim=-1)
    attn = (q @ k.transpose(-2, -1)) * temperature
    attn = attn.softmax(dim=-1)
    return (attn @ v).reshape(b, c, h, w)


# --------------------------------------------------------------------------- feature extraction
def _udsa(sd, p: str, side: Tensor) -> Tensor:
    """Prior U-net ``side_to_feaoneUDSA_2`` (arch.py:1815-1875)."""
    t = _lrelu(_conv(sd, p + "body.0", side, 1, 1))
    t = _lrelu(_conv(sd, p + "body.2", t, 2, 2))
    t = _lrelu(_conv(sd, p + "body.4", t, 2, 2))
    pooled = torch.cat([t.max(1, keepdim=True)[0], t.mean(1, keepdim=True)], 1)
    t = t * torch.sigmoid(_conv(sd, p + "body.6.spatial", pooled, 1, 3))
    t = _lrelu(_convT(sd, p + "body.7", t, 2, 2))
    t = _lrelu(_convT(sd, p + "body.9", t, 2, 2, 1))
    return _lrelu(_conv(sd, p + "body.11", t, 1, 1))


def _mdta(sd, p: str, x: Tensor) -> Tensor:
    qkv = _conv(sd, p + "qkv", x)
    qkv = F.conv2d(qkv, sd[p + "qkv_dwconv.weight"], None, padding=1, groups=qkv.shape[1])
    q, k, v = qkv.chunk(3, dim=1)
    out = _channel_attention(q, k, v, 8, sd[p + "temperature"])
    return _conv(sd, p + "project_out", out)


def feature_extraction(sd, x1: Tensor, x2: Tensor) -> Tensor:
    """``PAItransformerSA_2`` = three weight-shared rounds (arch.py:1451-1475)."""
    p = "transformer_feature_extraction.path1."
    for rnd in range(3):
        u = _udsa(sd, p + "side_to_feaoneUDSA.", x2)
        x2 = u + (x1 if rnd == 0 else x2)
        x1 = x1 + _mdta(sd, p + "attn.", _layernorm_c(sd, p + "norm1", x1))
        x1 = x1 + _conv(sd, p + "conv", _layernorm_c(sd, p + "norm2", x1), 1, 1) + x2
    return x1


# --------------------------------------------------------------------------- prior-fusion attention
def gumbel_hard_mask(v: Tensor, u: Tensor) -> Tensor:
    """softmax_c(v + G), G = -log(-log u), then hard threshold at 0.5 (arch.py:2168-2195)."""
    g = -(-u.log()).log()
    r = (v + g).softmax(1)
    return (r >= 0.5).to(v.dtype)


def rdab(sd, res: Tensor, x: Tensor, u: Tensor) -> Tensor:
    """``LLongRangAttention.forward`` (arch.py:2179-2249)."""
    p = "RDAB."
    b, c, h, w = x.shape
    t = F.relu(_conv(sd, p + "conv_du_re.0", res))
    t = F.relu(_conv(sd, p + "conv_du_re.2", t, 2, 2))
    t = t.mean((2, 3), keepdim=True)
    t = F.relu(_conv(sd, p + "conv_du_re2.0", t))
    v_max = t.expand(b, c, h, w)  # bilinear interpolation of a 1x1 map == broadcast
    mask = gumbel_hard_mask(v_max, u)
    inv = 1.0 - mask

    x_ = _conv(sd, p + "input_conv", x)
    q4, v4 = x_[:, :c], x_[:, c:]                     # 'b (qv c) h w', qv outer

    wW, bW = sd[p + "directW1_conv.weight"], sd[p + "directW1_conv.bias"]
    wH, bH = sd[p + "directH1_conv.weight"], sd[p + "directH1_conv.bias"]

    # row attention: sequences over W, features = channels; the (1,9) kernel slides over channels
    def rows(z):  # [b,c,h,w] -> [(b h), w, c]
        return z.permute(0, 2, 3, 1).reshape(b * h, w, c)

    sq = rows(mask * q4)
    sq = F.conv2d(sq.unsqueeze(1), wW, bW, padding=(0, 4)).squeeze(1)
    vv = F.conv2d(rows(v4).unsqueeze(1), wW, bW, padding=(0, 4)).squeeze(1)
    a = (sq @ sq.transpose(-2, -1)).softmax(-1)
    vv = a @ vv                                        # [(b h), w, c]

    # column attention: sequences over H; the (9,1) kernel slides over H
    def cols(z):  # [(b h), w, c] -> [(b w), h, c]
        return z.reshape(b, h, w, c).permute(0, 2, 1, 3).reshape(b * w, h, c)

    qc = F.conv2d(cols(sq).unsqueeze(1), wH, bH, padding=(4, 0)).squeeze(1)
    a = (qc @ qc.transpose(-2, -1)).softmax(-1)
    long_out = a @ cols(vv)                            # [(b w), h, c]
    long_out = long_out.reshape(b, w, h, c).permute(0, 3, 2, 1)

    # 8x8 window attention with the inverted mask
    ws = 8

    def wins(z):  # [b,c,h,w] -> [(b nh nw), 64, c]
        z = z.reshape(b, c, h // ws, ws, w // ws, ws).permute(0, 2, 4, 3, 5, 1)
        return z.reshape(b * (h // ws) * (w // ws), ws * ws, c)

    sq = wins(inv) * wins(q4)
    a = (sq @ sq.transpose(-2, -1)).softmax(-1)
    loc = a @ wins(v4)
    loc = loc.reshape(b, h // ws, w // ws, ws, ws, c).permute(0, 5, 1, 3, 2, 4).reshape(b, c, h, w)

    return _conv(sd, p + "fuse", torch.cat([long_out, loc], 1)) + x


# --------------------------------------------------------------------------- alignment
def flow_warp(x: Tensor, flow: Tensor) -> Tensor:
    """Bilinear sample of x at (col + flow[...,0], row + flow[...,1]); zeros outside (arch.py:3068-3099)."""
    _, _, h, w = x.shape
    gy, gx = torch.meshgrid(torch.arange(h, dtype=x.dtype), torch.arange(w, dtype=x.dtype), indexing="ij")
    vx = 2.0 * (gx + flow[..., 0]) / max(w - 1, 1) - 1.0
    vy = 2.0 * (gy + flow[..., 1]) / max(h - 1, 1) - 1.0
    return F.grid_sample(x, torch.stack((vx, vy), 3), mode="bilinear", padding_mode="zeros", align_corners=True)


def _resblock(sd, p, x):
    return x + _conv(sd, p + "conv2", F.relu(_conv(sd, p + "conv1", x, 1, 1)), 1, 1)


def dual_att_alignment(sd, x: Tensor, extra: Tensor, pred: Tensor, mv: Tensor) -> Tensor:
    """Live ``DualAttAlignment.forward`` (arch.py:3455-3496)."""
    p = "MV_deform_align."

    def gate(z):  # conv_du(avg_pool(z))
        y = z.mean((2, 3), keepdim=True)
        y = F.relu(_conv(sd, p + "conv_du.0", y))
        return torch.sigmoid(_conv(sd, p + "conv_du.2", y))

    def fusion_out(z):
        return F.relu(_conv(sd, p + "fusion_out.0", z))

    warped = flow_warp(extra, mv.permute(0, 2, 3, 1))
    k = fusion_out(torch.cat([warped, pred], 1))
    temp = sd[p + "temperature"]
    o1 = _conv(sd, p + "project_out", _channel_attention(x, k, warped * gate(warped), 4, temp))
    o2 = _conv(sd, p + "project_out", _channel_attention(x, k, pred * gate(pred), 4, temp))
    out = fusion_out(torch.cat([o1 + o2, x], 1))
    y = out.mean((2, 3), keepdim=True)
    y = torch.sigmoid(_conv(sd, p + "CALayer.conv_du.2", F.relu(_conv(sd, p + "CALayer.conv_du.0", y))))
    out = out * y
    out = _resblock(sd, p + "ResidualBlock1.", _resblock(sd, p + "ResidualBlock.", out))
    return out + x


# --------------------------------------------------------------------------- reconstruction trunk
def _half(x):
    return F.interpolate(x, scale_factor=0.5, mode="bilinear", align_corners=False)


def _double(x):
    return F.interpolate(x, scale_factor=2.0, mode="bilinear", align_corners=False)


def block_(sd, p: str, x: Tensor) -> Tensor:
    def body(z):
        return _conv(sd, p + "body.2", _lrelu(_conv(sd, p + "body.0", z, 1, 1)), 1, 1)

    def down(z):
        return _half(_conv(sd, p + "down.0", z))

    def up(z):
        return _double(_conv(sd, p + "up.0", z))

    return x + body(x) + up(body(down(x))) + down(body(up(x)))


def recon_trunk(sd, x: Tensor, groups: int = 7, blocks: int = 3) -> Tensor:
    y = x
    for g in range(groups):
        p = f"recon_trunk.body.{g}."
        r = y
        for bidx in range(blocks):
            r = block_(sd, p + f"body.{bidx}.", r)
        y = y + _conv(sd, p + "conv", r, 1, 1)
    return y + x


# --------------------------------------------------------------------------- whole forward
def cvsr_v8_forward(sd: Dict[str, Tensor], x: Tensor, mvs0: Optional[Tensor], mvs1: Tensor, pms: Tensor,
                    rms: Tensor, ufs: Tensor, pre_L1_fea: Optional[Tensor] = None,
                    gumbel_u: Optional[List[Tensor]] = None, taps: Optional[dict] = None):
    """Returns ``(out [B,1,4H,4W], L1_fea [B*7,64,H,W])`` like arch.py:4406-4481.  ``mvs0`` is ignored
    by the reference as well (only ``mvs1`` is read, arch.py:4445)."""
    B, N, C, H, W = x.shape
    ctr = N // 2
    x_center = x[:, ctr]
    if pre_L1_fea is None:
        f = _lrelu(_conv(sd, "conv_first", x.reshape(-1, C, H, W), 1, 1))
        s = _conv(sd, "conv_second", pms.reshape(-1, C, H, W), 1, 1)
        L1 = feature_extraction(sd, f, s)
    else:
        f = _lrelu(_conv(sd, "conv_first", x[:, -1], 1, 1))
        s = _conv(sd, "conv_second", pms[:, -1], 1, 1)
        new = feature_extraction(sd, f, s).unsqueeze(1)
        L1 = torch.cat([pre_L1_fea.view(B, N, -1, H, W)[:, 1:], new], 1).reshape(B * N, -1, H, W)
    if taps is not None:
        taps["L1_fea"] = L1
    fea = L1.view(B, N, -1, H, W)
    if ufs.shape[1] != 1:                      # [B,7,1,H,W] layout -> [B,1,7,H,W]
        ufs = ufs.transpose(1, 2)
        rms = rms.transpose(1, 2)

    aligned = []
    draw = 0
    for i in range(N):
        if i == ctr:
            aligned.append(fea[:, i])
            continue
        ufs_prior = _conv(sd, "conv_expand_ufs", ufs[:, :, i], 1, 1)
        rms_prior = _conv(sd, "conv_expand_rms", rms[:, :, i], 1, 1)
        fea_com = fea[:, i] + rms_prior
        if gumbel_u is None:
            u = torch.rand_like(rms_prior)
            while bool((u == 0).any()):
                u = torch.rand_like(rms_prior)
        else:
            u = gumbel_u[draw]
        draw += 1
        x_n = rdab(sd, rms_prior, fea_com, u)
        fea_i = _conv(sd, "conv_expand_fea_r", torch.cat([fea[:, i], x_n], 1), 1, 1)
        al = dual_att_alignment(sd, fea[:, ctr], fea_i, ufs_prior, mvs1[:, i])
        if taps is not None:
            taps[f"rdab_{i}"] = x_n
            taps[f"align_{i}"] = al
        aligned.append(al)

    fused = _lrelu(_conv(sd, "tsa_fusion", torch.stack(aligned, 1).reshape(B, -1, H, W)))
    trunk = recon_trunk(sd, fused)
    if taps is not None:
        taps["fused"] = fused
        taps["trunk"] = trunk
    out = _lrelu(F.pixel_shuffle(_conv(sd, "upconv1", trunk), 2))
    out = _lrelu(F.pixel_shuffle(_conv(sd, "upconv2", out), 2))
    out = _conv(sd, "conv_last", out, 1, 1)
    out = out + F.interpolate(x_center, scale_factor=4.0, mode="bilinear", align_corners=False)
    return out, L1


# --------------------------------------------------------------------------- deterministic weights / inputs
def state_dict_spec() -> List[tuple]:
    """(key, shape, fan_in-or-None, scale) for all 261 entries of the reference ``CVSR_V8().state_dict()``
    (SURVEY section 8b).  Init mirrors the reference's: default Conv2d init except the 0.1-scaled Kaiming
    blocks (arch.py:275-291) -- close enough for well-conditioned activations; parity tests never depend on
    the init law because the same tensors are loaded on both sides."""
    spec: List[tuple] = []

    def conv(key, co, ci, k, bias=True, kh=None, kw=None, mode="default"):
        kh = kh or k
        kw = kw or k
        spec.append((key + ".weight", (co, ci, kh, kw), ci * kh * kw, mode))
        if bias:
            spec.append((key + ".bias", (co,), ci * kh * kw, "bias0" if mode == "kaiming0.1" else "bias"))

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
    u = p + "side_to_feaoneUDSA.body."
    conv(u + "0", 16, 64, 3)
    conv(u + "2", 16, 16, 3)
    conv(u + "4", 16, 16, 3)
    conv(u + "6.spatial", 1, 2, 7)
    conv(u + "7", 16, 16, 3)      # ConvTranspose2d weight is [in, out, k, k]; square here
    conv(u + "9", 16, 16, 3)
    conv(u + "11", 64, 16, 3)
    conv("conv_expand_fea_r", 64, 128, 3)
    conv("conv_expand_ufs", 64, 1, 3)
    conv("conv_expand_rms", 64, 1, 3)
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
    conv("upconv1", 256, 64, 1)
    conv("upconv2", 256, 64, 1)
    conv("conv_last", 1, 64, 3)
    a = "MV_deform_align."
    spec.append((a + "temperature", (4, 1, 1), None, "ones"))
    conv(a + "conv_du.0", 4, 64, 1)
    conv(a + "conv_du.2", 64, 4, 1)
    conv(a + "project_out", 64, 64, 1, bias=False)
    conv(a + "fusion_in.0", 64, 128, 1)
    conv(a + "fusion_in.2", 64, 64, 1)
    conv(a + "fusion_out.0", 64, 128, 1, bias=False)
    conv(a + "CALayer.conv_du.0", 64, 64, 1)
    conv(a + "CALayer.conv_du.2", 64, 64, 1)
    for rb in ("ResidualBlock.", "ResidualBlock1."):
        conv(a + rb + "conv1", 64, 64, 3, mode="kaiming0.1")
        conv(a + rb + "conv2", 64, 64, 3, mode="kaiming0.1")
    r = "RDAB."
    conv(r + "input_conv", 128, 64, 1)
    conv(r + "conv_du_re.0", 64, 64, 1)
    conv(r + "conv_du_re.2", 64, 64, 3)
    conv(r + "conv_du_re2.0", 64, 64, 1)
    conv(r + "fuse", 64, 128, 1)
    conv(r + "directW1_conv", 1, 1, 1, kh=1, kw=9)
    conv(r + "directH1_conv", 1, 1, 1, kh=9, kw=1)
    return spec


def make_state_dict(seed: int = 0, perturb: bool = True) -> Dict[str, Tensor]:
    """Deterministic weights from ``numpy.random.RandomState`` (bit-stable across numpy versions and hosts).

    ``perturb`` moves LayerNorm affine / temperatures / zero biases off their trivial init so that parity
    tests exercise them."""
    rs = np.random.RandomState(seed)
    sd: Dict[str, Tensor] = {}
    for key, shape, fan_in, mode in state_dict_spec():
        if mode == "default":
            bound = 1.0 / math.sqrt(fan_in)            # kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), ..)
            a = rs.uniform(-bound, bound, size=shape)
        elif mode == "kaiming0.1":
            a = rs.standard_normal(size=shape) * (0.1 * math.sqrt(2.0 / fan_in))
        elif mode == "bias":
            bound = 1.0 / math.sqrt(fan_in)
            a = rs.uniform(-bound, bound, size=shape)
        elif mode == "bias0":
            a = rs.uniform(-0.02, 0.02, size=shape) if perturb else np.zeros(shape)
        elif mode == "ones":
            a = 1.0 + (rs.uniform(-0.2, 0.2, size=shape) if perturb else 0.0) * np.ones(shape)
        elif mode == "zeros":
            a = rs.uniform(-0.1, 0.1, size=shape) if perturb else np.zeros(shape)
        else:
            raise ValueError(mode)
        sd[key] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return sd


def make_inputs(B: int, H: int, W: int, seed: int, layout: str = "b1n", pad_rows: int = 0):
    """Synthetic clip with the statistics of SURVEY section 8(d).  Returns dict of CPU fp32 tensors:
    x, mvs0, mvs1, pms, rms, ufs, gumbel_u (list of 6)."""
    rs = np.random.RandomState(seed)
    N = NFRAMES

    def u8(shape):
        return rs.randint(0, 256, size=shape).astype(np.float32) / 255.0

    x = u8((B, N, 1, H, W))
    if pad_rows:
        x[..., H - pad_rows:, :] = 0.0
    pms = u8((B, N, 1, H, W))
    ufs = u8((B, 1, N, H, W))
    rms = np.clip(np.round(rs.standard_normal((B, 1, N, H, W)) * 6.0), -128, 127).astype(np.float32) / 255.0
    hb, wb = (H + 7) // 8, (W + 7) // 8
    m = rs.randint(-64, 64, size=(B, 2, hb, wb)).astype(np.float32)
    base = np.repeat(np.repeat(m, 8, axis=2), 8, axis=3)[:, :, :H, :W] / 128.0
    scale = np.array([3, 2, 1, 0, -1, -2, -3], dtype=np.float32).reshape(1, N, 1, 1, 1)
    mvs1 = base[:, None] * scale
    mvs0 = np.zeros_like(mvs1)
    gum = []
    for _ in range(N - 1):
        u = rs.random_sample((B, NF, H, W)).astype(np.float32)
        u[u == 0] = 0.5                                  # the reference redraws on exact zeros
        gum.append(torch.from_numpy(u))
    if layout == "bn1":
        ufs = np.ascontiguousarray(ufs.transpose(0, 2, 1, 3, 4))
        rms = np.ascontiguousarray(rms.transpose(0, 2, 1, 3, 4))
    t = torch.from_numpy
    return dict(x=t(x), mvs0=t(mvs0), mvs1=t(np.ascontiguousarray(mvs1)), pms=t(pms), rms=t(rms), ufs=t(ufs),
                gumbel_u=gum)


def psnr_y(a: Tensor, b: Tensor, crop_border: int = 4) -> float:
    """PSNR on the 0-255 scale, border-cropped, float64 (metric/psnr_ssim.py:278-317 semantics)."""
    a = (a.clamp(0, 1) * 255.0).double()
    b = (b.clamp(0, 1) * 255.0).double()
    if crop_border:
        a = a[..., crop_border:-crop_border, crop_border:-crop_border]
        b = b[..., crop_border:-crop_border, crop_border:-crop_border]
    mse = ((a - b) ** 2).mean().item()
    return float("inf") if mse == 0 else 20.0 * math.log10(255.0 / math.sqrt(mse))
