"""CPU checks of the host-side algebra behind two kernels (no GPU, no HIP library): the composed stencils are evaluated with
plain torch and compared with the layer-by-layer convolutions of the reference (arch/SIDECVSR_our.py:4420, 1819-1820)."""
import torch
import torch.nn.functional as F


def _udsa_head_formula(img, wc, bt, b0):
    """What cdfo_udsa_head computes: out[o](p) = b0[o] + sum_t [p + t inside] (bt[t][o] + sum_u wc[t][u][o] img(p + t + u)), LeakyReLU 0.1."""
    B, _, H, W = img.shape
    pad = F.pad(img[:, 0], (2, 2, 2, 2))                                   # zero-padded image, offset 2
    inside = F.pad(torch.ones(H, W, dtype=img.dtype), (1, 1, 1, 1))        # 1 inside the image, offset 1
    out = b0.view(1, 16, 1, 1).expand(B, 16, H, W).clone()
    for ty in range(3):
        for tx in range(3):
            t = ty * 3 + tx
            m = inside[ty:ty + H, tx:tx + W]                               # is p + t inside?
            acc = bt[t].view(1, 16, 1, 1).expand(B, 16, H, W).clone()
            for uy in range(3):
                for ux in range(3):
                    v = pad[:, ty + uy:ty + uy + H, tx + ux:tx + ux + W]   # img(p + t + u)
                    acc = acc + wc[t, uy * 3 + ux].view(1, 16, 1, 1) * v.unsqueeze(1)
            out = out + m * acc
    return F.leaky_relu(out, 0.1)


def test_udsa_head_composition_matches_the_two_convolutions_including_borders():
    # pack_udsa_head is pure torch; cdfo_amd.kernels loads the HIP library lazily, so importing it needs no GPU
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(3)
    img = torch.randn(2, 1, 9, 13, generator=g, dtype=torch.float64)
    w2, b2 = torch.randn(64, 1, 3, 3, generator=g, dtype=torch.float64) / 3, torch.randn(64, generator=g, dtype=torch.float64)
    w0, b0 = torch.randn(16, 64, 3, 3, generator=g, dtype=torch.float64) / 24, torch.randn(16, generator=g, dtype=torch.float64)
    wc, bt, b0p = K.pack_udsa_head(w0, b0, w2, b2)
    ref = F.leaky_relu(F.conv2d(F.conv2d(img, w2, b2, padding=1), w0, b0, padding=1), 0.1)
    got = _udsa_head_formula(img, wc.double(), bt.double(), b0p.double())
    assert (got - ref).abs().max().item() <= 1e-5        # fp32 packing of fp64-composed weights


def test_one_by_one_after_three_by_three_composes_exactly():
    """up.0 o body.2 (Block_'s half-resolution branch) and conv_du_re.0 o conv_expand_rms: W'[o][c][t] = sum_m W1[o][m] W3[m][c][t],
    b' = W1 b3 + b1 -- the einsum used in cdfo_amd/cvsr_v8.py::_weights."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 8, 7, 9, generator=g, dtype=torch.float64)
    w3, b3 = torch.randn(6, 8, 3, 3, generator=g, dtype=torch.float64), torch.randn(6, generator=g, dtype=torch.float64)
    w1, b1 = torch.randn(5, 6, generator=g, dtype=torch.float64), torch.randn(5, generator=g, dtype=torch.float64)
    ref = F.conv2d(F.conv2d(x, w3, b3, padding=1), w1.view(5, 6, 1, 1), b1)
    wc = torch.einsum("po,ocyx->pcyx", w1, w3)
    got = F.conv2d(x, wc, w1 @ b3 + b1, padding=1)
    assert (got - ref).abs().max().item() <= 1e-10


def test_stride_two_convolution_as_tap_masked_convolution_over_space_to_depth():
    """conv_du_re.2 (3x3, stride 2, pad 2) = a stride-1 pad-1 convolution over the space-to-depth tensor [H/2+1, W/2+1, 4C] (last row
    / column zero) with weights on the taps (-1, 0) x (-1, 0) only -- the mapping built in cdfo_amd/cvsr_v8.py::_weights."""
    g = torch.Generator().manual_seed(9)
    Cc, H, W = 3, 8, 12
    x = torch.randn(2, Cc, H, W, generator=g, dtype=torch.float64)
    w, b = torch.randn(5, Cc, 3, 3, generator=g, dtype=torch.float64), torch.randn(5, generator=g, dtype=torch.float64)
    ref = F.conv2d(x, w, b, stride=2, padding=2)
    s2d = torch.zeros(2, 4 * Cc, H // 2 + 1, W // 2 + 1, dtype=torch.float64)
    for a_ in range(2):
        for b_ in range(2):
            s2d[:, (a_ * 2 + b_) * Cc:(a_ * 2 + b_ + 1) * Cc, :H // 2, :W // 2] = x[:, :, a_::2, b_::2]
    ws = torch.zeros(5, 4, Cc, 3, 3, dtype=torch.float64)
    for a_ in range(2):
        for b_ in range(2):
            for ty in range(2):
                for tx in range(2):
                    dy, dx = 2 * ty + a_, 2 * tx + b_
                    if dy <= 2 and dx <= 2:
                        ws[:, a_ * 2 + b_, :, ty, tx] = w[:, :, dy, dx]
    got = F.conv2d(s2d, ws.view(5, 4 * Cc, 3, 3), b, padding=1)
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() <= 1e-10


def test_offset_head_fallback_selection_by_frame_size():
    """the weights-stationary offset head addresses one image's offset planes with 32-bit offsets; a 1920x1080 frame at
    dg = 16 (third = 144) exceeds them and must fall through to the tiled head instead of raising CDFO_EINVAL."""
    from cdfo_amd import kernels as K
    assert K.conv_offset_mask_ws_fits(8, 272, 480, 144)
    assert K.conv_offset_mask_ws_fits(1, 544, 960, 144)
    assert not K.conv_offset_mask_ws_fits(1, 1080, 1920, 144)          # 1080*1920*288*4 = 2.39e9 >= 2^31
    assert not K.conv_offset_mask_ws_fits(64, 544, 960, 144)           # the fp16 source of the whole batch >= 2 GiB
