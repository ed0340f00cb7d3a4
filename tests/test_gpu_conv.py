"""GPU parity: cdfo_conv_igemm (through the C-ABI) against torch-cpu F.conv2d on the same seeded inputs."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _cmp(got_nhwc, ref_nchw, tol, what):
    got = got_nhwc.permute(0, 3, 1, 2).cpu()
    err = (got - ref_nchw).abs().max().item()
    scale = ref_nchw.abs().max().item()
    assert err <= tol * max(1.0, scale), f"{what}: max-abs {err} (ref scale {scale})"


CASES = [
    # Cin, Cout, ks, stride, pad, H, W, B
    (64, 64, 3, 1, 1, 16, 16, 2),
    (64, 256, 3, 1, 1, 24, 40, 1),
    (256, 64, 3, 1, 1, 20, 18, 1),
    (64, 16, 3, 1, 1, 17, 33, 2),
    (16, 64, 3, 1, 1, 8, 8, 1),
    (64, 192, 1, 1, 0, 16, 24, 2),
    (448, 64, 1, 1, 0, 9, 21, 1),
    (64, 64, 3, 2, 2, 16, 24, 2),
    (64, 64, 3, 2, 2, 17, 19, 1),
]


@pytest.mark.parametrize("Cin,Cout,ks,stride,pad,H,W,B", CASES)
def test_conv_igemm_matches_torch(Cin, Cout, ks, stride, pad, H, W, B):
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(Cin * 7 + Cout + ks + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.leaky_relu(F.conv2d(x, w, b, stride=stride, padding=pad), 0.1)
    pc = K.pack_conv(w.cuda(), b.cuda())
    out = K.conv([_nhwc(x).cuda()], pc, stride=stride, pad=pad, act=K.ACT_LRELU)
    torch.cuda.synchronize()
    _cmp(out, ref, 2e-5, "conv")


def test_conv_concat_residual_and_slices():
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(5)
    B, H, W = 2, 12, 20
    x0 = torch.randn(B, 64, H, W, generator=g)
    x1 = torch.randn(B, 64, H, W, generator=g)
    r1 = torch.randn(B, 64, H, W, generator=g)
    r2 = torch.randn(B, 64, H, W, generator=g)
    w = torch.randn(64, 128, 3, 3, generator=g) / 34.0
    b = torch.randn(64, generator=g)
    ref = F.relu(F.conv2d(torch.cat([x0, x1], 1), w, b, padding=1)) + r1 + r2
    wide = torch.zeros(B, H, W, 192, device="cuda")
    wide[..., 64:128] = _nhwc(x1).cuda()                    # second source lives in a channel slice
    outbuf = torch.zeros(B, H, W, 128, device="cuda")
    pc = K.pack_conv(w.cuda(), b.cuda())
    K.conv([_nhwc(x0).cuda(), wide[..., 64:128]], pc, pad=1, act=K.ACT_RELU, res1=_nhwc(r1).cuda(),
           res2=_nhwc(r2).cuda(), out=outbuf[..., 64:128])
    torch.cuda.synchronize()
    _cmp(outbuf[..., 64:128], ref, 2e-5, "concat+res")
    assert outbuf[..., :64].abs().max().item() == 0.0


def test_conv_pixel_shuffle_store():
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 64, 10, 14, generator=g)
    w = torch.randn(256, 64, 1, 1, generator=g) / 8.0
    b = torch.randn(256, generator=g)
    ref = F.leaky_relu(F.pixel_shuffle(F.conv2d(x, w, b), 2), 0.1)
    pc = K.pack_conv(w.cuda(), b.cuda(), shuffle2=True)
    out = K.conv([_nhwc(x).cuda()], pc, act=K.ACT_LRELU)
    torch.cuda.synchronize()
    assert tuple(out.shape) == (1, 20, 28, 64)
    _cmp(out, ref, 2e-5, "shuffle")


def test_conv_per_image_weights():
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(7)
    B = 3
    x = torch.randn(B, 64, 8, 16, generator=g)
    ws = torch.randn(B, 64, 64, 1, 1, generator=g) / 8.0
    ref = torch.cat([F.conv2d(x[i:i + 1], ws[i]) for i in range(B)], 0)
    pcs = [K.pack_conv(ws[i].cuda(), None) for i in range(B)]
    pc = pcs[0]
    pc.w = torch.stack([p.w for p in pcs], 0).contiguous()
    pc.w_bstride = pc.w.stride(0)
    out = K.conv([_nhwc(x).cuda()], pc)
    torch.cuda.synchronize()
    _cmp(out, ref, 2e-5, "per-image weights")


def test_layout_roundtrip():
    from cdfo_amd import kernels as K
    x = torch.randn(2, 64, 9, 13)
    y = K.nchw_to_nhwc(x.cuda())
    assert torch.equal(y.cpu(), x.permute(0, 2, 3, 1).contiguous())
    z = K.nhwc_to_nchw(y)
    assert torch.equal(z.cpu(), x)


@pytest.mark.parametrize("Cin,Cout,H,W,B", [(64, 256, 24, 40, 1), (256, 64, 20, 18, 2), (128, 64, 16, 16, 1), (64, 64, 33, 17, 1)])
@pytest.mark.parametrize("prec,tol", [(1, 3e-5), (2, 2e-2), (3, 1e-3)])
def test_conv3x3_bf16_family(Cin, Cout, H, W, B, prec, tol):
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(Cin + Cout + H + prec)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    r1 = torch.randn(B, Cout, H, W, generator=g)
    ref = F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.1) + r1
    pc = K.pack_conv(w.cuda(), b.cuda())
    assert pc.wq is not None
    if Cin == 128:
        xs = _nhwc(x).cuda()
        srcs = [xs[..., :64].contiguous(), xs[..., 64:]]          # two sources, the second a strided slice
    else:
        srcs = [_nhwc(x).cuda()]
    out = K.conv(srcs, pc, pad=1, act=K.ACT_LRELU, res1=_nhwc(r1).cuda(), prec=prec)
    torch.cuda.synchronize()
    _cmp(out, ref, tol, f"conv3x3 prec={prec}")


def test_conv1x1_fused_layernorm_and_wide_sources():
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(11)
    B, H, W = 2, 24, 40
    x = torch.randn(B, 64, H, W, generator=g) * 3 + 1
    gamma, beta = torch.randn(64, generator=g), torch.randn(64, generator=g)
    w = torch.randn(192, 64, 1, 1, generator=g) / 8.0
    mu = x.mean(1, keepdim=True)
    var = x.var(1, keepdim=True, unbiased=False)
    ln = (x - mu) / torch.sqrt(var + 1e-5) * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)
    ref = F.conv2d(ln, w)
    pc = K.pack_conv(w.cuda(), None)
    out = K.conv([_nhwc(x).cuda()], pc, ln=(gamma.cuda(), beta.cuda()))
    torch.cuda.synchronize()
    _cmp(out, ref, 3e-5, "ln+1x1")
    # 7 x 64-channel sources (tsa_fusion shape) through the 64-channel-chunk path
    xs = [torch.randn(1, 64, 9, 21, generator=g) for _ in range(7)]
    w2 = torch.randn(64, 448, 1, 1, generator=g) / 21.0
    b2 = torch.randn(64, generator=g)
    ref2 = F.leaky_relu(F.conv2d(torch.cat(xs, 1), w2, b2), 0.1)
    out2 = K.conv([_nhwc(t).cuda() for t in xs], K.pack_conv(w2.cuda(), b2.cuda()), act=K.ACT_LRELU)
    torch.cuda.synchronize()
    _cmp(out2, ref2, 2e-5, "7-source 1x1")


def test_conv3x3_fp16_intermediate_chain():
    """conv1 (fp16x2, fp16 output, optionally space-to-depth) -> conv2 (fp16 source, 1 pass) against torch on the
    fp16-rounded intermediate."""
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(21)
    B, H, W = 1, 16, 40
    x = torch.randn(B, 64, H, W, generator=g)
    w1 = torch.randn(256, 64, 3, 3, generator=g) / 24.0
    b1 = torch.randn(256, generator=g) * 0.1
    w2 = torch.randn(64, 256, 3, 3, generator=g) / 48.0
    b2 = torch.randn(64, generator=g) * 0.1
    r = torch.randn(B, 64, H, W, generator=g)
    pc1, pc2 = K.pack_conv(w1.cuda(), b1.cuda()), K.pack_conv(w2.cuda(), b2.cuda())
    t = K.conv([_nhwc(x).cuda()], pc1, pad=1, act=K.ACT_LRELU, prec=K.PREC_FP16X2, out_f16=True)
    assert t.dtype == torch.float16
    t_ref = F.leaky_relu(F.conv2d(x, w1, b1, padding=1), 0.1)
    _cmp(t.float(), t_ref, 2e-3, "conv1 -> fp16")
    out = K.conv([t], pc2, pad=1, res1=_nhwc(r).cuda())
    torch.cuda.synchronize()
    ref = F.conv2d(t.float().permute(0, 3, 1, 2).cpu(), w2.half().float(), b2, padding=1) + r
    _cmp(out, ref, 1e-4, "conv2 from fp16 source")
    # space-to-depth + fp16 store
    t2 = K.conv([_nhwc(x).cuda()], pc1, pad=1, act=K.ACT_LRELU, prec=K.PREC_FP16X2, s2d=True, out_f16=True)
    torch.cuda.synchronize()
    exp = t.view(B, H // 2, 2, W // 2, 2, 256).permute(0, 1, 3, 2, 4, 5).reshape(B, H // 2, W // 2, 1024)
    assert torch.equal(t2, exp)


@pytest.mark.parametrize("srcs,Cout,res,shuf", [([64], 192, False, False), ([64], 64, True, False), ([64, 64, 64], 64, False, False),
                                                ([128], 64, True, False), ([64] * 7, 64, False, False), ([64], 256, False, True),
                                                ([64], 128, False, False)])
def test_conv1x1_bf16x3(srcs, Cout, res, shuf):
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(sum(srcs) + Cout)
    B, H, W = 2, 13, 21                       # 273 pixels: partial last tile
    xs = [torch.randn(B, c, H, W, generator=g) for c in srcs]
    cin = sum(srcs)
    w = torch.randn(Cout, cin, 1, 1, generator=g) / cin ** 0.5
    b = torch.randn(Cout, generator=g)
    y = F.conv2d(torch.cat(xs, 1), w, b)
    if shuf:
        ref = F.leaky_relu(F.pixel_shuffle(y, 2), 0.1)
    else:
        ref = F.leaky_relu(y, 0.1)
    r = torch.randn(B, Cout, H, W, generator=g) if res else None
    if res:
        ref = ref + r
    pc = K.pack_conv(w.cuda(), b.cuda(), shuffle2=shuf)
    out = K.conv([_nhwc(t).cuda() for t in xs], pc, act=K.ACT_LRELU, res1=None if r is None else _nhwc(r).cuda(),
                 prec=K.PREC_BF16X3)
    torch.cuda.synchronize()
    _cmp(out, ref, 3e-5, "conv1x1 bf16x3")


def test_conv1x1_bf16x3_layernorm_and_per_image_weights():
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(77)
    B, H, W = 3, 8, 24
    x = torch.randn(B, 64, H, W, generator=g) * 2 + 0.5
    gamma, beta = torch.randn(64, generator=g), torch.randn(64, generator=g)
    w = torch.randn(192, 64, 1, 1, generator=g) / 8.0
    mu = x.mean(1, keepdim=True)
    var = x.var(1, keepdim=True, unbiased=False)
    ln = (x - mu) / torch.sqrt(var + 1e-5) * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)
    out = K.conv([_nhwc(x).cuda()], K.pack_conv(w.cuda(), None), ln=(gamma.cuda(), beta.cuda()), prec=K.PREC_BF16X3)
    torch.cuda.synchronize()
    _cmp(out, F.conv2d(ln, w), 5e-5, "ln + 1x1 bf16x3")
    ws = torch.randn(B, 64, 64, 1, 1, generator=g) / 8.0
    ref = torch.cat([F.conv2d(x[i:i + 1], ws[i]) for i in range(B)], 0)
    pcs = [K.pack_conv(ws[i].cuda(), None) for i in range(B)]
    pc = pcs[0]
    pc.w = torch.stack([p.w for p in pcs], 0).contiguous()
    pc.w_bstride = pc.w.stride(0)
    out = K.conv([_nhwc(x).cuda()], pc, prec=K.PREC_BF16X3)
    torch.cuda.synchronize()
    _cmp(out, ref, 5e-5, "per-image weights bf16x3")


WS_CASES = [
    # Cout, H, W, B, s2d, act      (W % 32 != 0 and tiny images exercise the masked tile edges; B*H/2*tiles_x both
    (256, 24, 40, 1, False, 1),    # below and above the number of wave streams)
    (64, 16, 64, 2, False, 0),
    (128, 6, 34, 3, False, 2),
    (256, 36, 52, 2, True, 1),
    (256, 272, 480, 2, True, 1),
    (64, 2, 8, 1, False, 1),       # a single tile row, narrower than a tile
    (64, 8, 200, 3, False, 0),
    (192, 20, 70, 2, False, 1),    # three 64-channel blocks: the XCD's 32 workgroup slots do not divide evenly
    (320, 34, 36, 1, True, 2),     # five blocks, a ragged last tile row (34 = 2 x 16 + 2), space-to-depth store
]


@pytest.mark.parametrize("Cout,H,W,B,s2d,act", WS_CASES)
def test_conv3x3_c64_ws(Cout, H, W, B, s2d, act):
    """Weights-stationary Block_.body[0] kernel vs torch-cpu conv2d on the SAME fp16-rounded operands (so the only
    differences are the fp32 accumulation order and the fp16 rounding of the result)."""
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(Cout + H + W)
    x = torch.randn(B, 64, H, W, generator=g)
    w = torch.randn(Cout, 64, 3, 3, generator=g) / 24.0
    b = torch.randn(Cout, generator=g)
    xh, wh = x.half().float(), w.half().float()
    ref = F.conv2d(xh, wh, b, padding=1)
    ref = {0: ref, 1: F.leaky_relu(ref, 0.1), 2: F.relu(ref)}[act]
    pc = K.pack_conv(w.cuda(), b.cuda())
    src = K.to_cp16(_nhwc(x).cuda())
    assert torch.equal(src.cpu().permute(0, 1, 4, 2, 3).reshape(B, 64, H, W), x.half())
    out = K.conv3x3_ws(src, pc, act=act, s2d=s2d)
    torch.cuda.synchronize()
    got = K.from_cp16(out).float().cpu()
    if s2d:   # [B,H/2,W/2,(py,px,c)] -> [B,H,W,c]
        got = got.view(B, H // 2, W // 2, 2, 2, Cout).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, Cout)
    got = got.permute(0, 3, 1, 2)
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= 1.5e-3 * max(1.0, scale), f"ws conv: max-abs {err} (ref scale {scale})"


WINO_CASES = [
    # Cout, H, W, B, s2d, act      (ragged strips, segments shorter than a batch, one-row images, several units per workgroup)
    (256, 24, 40, 1, False, 1),
    (128, 16, 64, 2, False, 0),
    (128, 6, 34, 3, False, 2),
    (256, 36, 52, 2, True, 1),
    (256, 272, 480, 2, True, 1),
    (128, 2, 8, 1, False, 1),
    (128, 1, 30, 2, False, 1),
    (256, 8, 200, 3, False, 0),
    (384, 20, 70, 2, True, 1),      # three 128-channel groups: the XCD's 32 workgroup slots do not divide evenly
    (256, 136, 240, 8, False, 1),   # the half-resolution launch of c3: half-empty last strip, 17-row segments
    (256, 48, 672, 13, True, 1),    # several units per workgroup (unit transitions inside the persistent loop), odd unit count
    (128, 44, 100, 30, False, 1),   # one 128-channel group: 256 workgroup lanes, ragged last strip, several units each
]


@pytest.mark.parametrize("Cout,H,W,B,s2d,act", WINO_CASES)
def test_conv3x3_c64_wino(Cout, H, W, B, s2d, act):
    """Row-streaming Winograd F(2,3) kernel (Block_.body[0], arch.py:383-387) vs torch-cpu conv2d in float64 on the fp16-rounded
    source and the UNROUNDED weights: the kernel rounds the transformed weights, not the weights.  Bound: the direct kernel's
    (test_conv3x3_c64_ws) -- the fp16 rounding of the result dominates both."""
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(Cout + H + W)
    x = torch.randn(B, 64, H, W, generator=g)
    w = torch.randn(Cout, 64, 3, 3, generator=g) / 24.0
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.half().double(), w.double(), b.double(), padding=1)
    ref = {0: ref, 1: F.leaky_relu(ref, 0.1), 2: F.relu(ref)}[act].float()
    pc = K.pack_conv(w.cuda(), b.cuda())
    assert pc.ww is not None
    src = K.to_cp16(_nhwc(x).cuda())
    out = K.conv3x3_wino(src, pc, act=act, s2d=s2d)
    torch.cuda.synchronize()
    got = K.from_cp16(out).float().cpu()
    if s2d:   # [B,H/2,W/2,(py,px,c)] -> [B,H,W,c]
        got = got.view(B, H // 2, W // 2, 2, 2, Cout).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, Cout)
    got = got.permute(0, 3, 1, 2)
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= 1.5e-3 * max(1.0, scale), f"wino conv: max-abs {err} (ref scale {scale})"
    # and against the direct weights-stationary kernel on the same operands: two fp16 roundings of nearly the same number
    if H % 2 == 0:
        direct = K.from_cp16(K.conv3x3_ws(src, pc, act=act, s2d=s2d)).float().cpu()
        if s2d:
            direct = direct.view(B, H // 2, W // 2, 2, 2, Cout).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, Cout)
        d = (got - direct.permute(0, 3, 1, 2)).abs().max().item()
        assert d <= 3e-3 * max(1.0, scale), f"wino vs direct: {d}"


def test_pack_conv3x3_wino_layout():
    """cdfo_pack_conv3x3_wino against the layout include/cdfo_hip.h documents (torch arithmetic in float64)."""
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(5)
    w = torch.randn(128, 64, 3, 3, generator=g)
    pc = K.pack_conv(w.cuda(), None)
    gg = w.double()
    U = torch.stack([gg[..., 0], (gg[..., 0] + gg[..., 1] + gg[..., 2]) / 2, (gg[..., 0] - gg[..., 1] + gg[..., 2]) / 2, gg[..., 2]], -1)
    # U [o, c, dy, xi] -> [cb, dy, xi, sc, kg, i, e]
    exp = U.view(8, 16, 2, 4, 8, 3, 4).permute(0, 5, 6, 2, 3, 1, 4).reshape(-1)
    got = pc.ww.cpu().double()
    assert (got - exp).abs().max().item() <= 2.0 ** -11 * exp.abs().max().item()


RING_CASES = [
    # Cin, Cout, H, W, B, res, out_f16
    (256, 64, 20, 40, 1, True, False),
    (256, 64, 16, 32, 3, False, False),
    (128, 128, 34, 70, 2, True, False),     # two output blocks, ragged tile edges in both directions
    (64, 64, 48, 96, 2, False, True),
    (256, 64, 272, 480, 2, True, False),    # more tiles than workgroups: ring runs across tile boundaries
    (64, 64, 8, 8, 1, True, False),         # smaller than one tile
    (32, 64, 18, 200, 3, True, False),      # two chunks only, ragged right edge
]


@pytest.mark.parametrize("Cin,Cout,H,W,B,res,out_f16", RING_CASES)
def test_conv3x3_ring_dense(Cin, Cout, H, W, B, res, out_f16):
    """LDS-DMA ring kernel vs torch-cpu conv2d on the same fp16-rounded operands."""
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(Cin + Cout + H + W)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    r1 = torch.randn(B, Cout, H, W, generator=g) if res else None
    ref = F.relu(F.conv2d(x.half().float(), w.half().float(), b, padding=1))
    if res:
        ref = ref + r1
    pc = K.pack_conv(w.cuda(), b.cuda())
    src = K.to_cp16(_nhwc(x).cuda())
    out = K.conv_ring(src, pc, act=K.ACT_RELU, res1=_nhwc(r1).cuda() if res else None, out_f16=out_f16)
    torch.cuda.synchronize()
    _cmp(out.float(), ref, 1.5e-3 if out_f16 else 1e-4, "ring conv")


def test_conv3x3_ring_sparse_taps_halfsplit_source():
    """cdfo_conv_args.src_halfsplit: the four-tap ring kernel on a source whose rows are [half][W][8] gives the natural layout's result."""
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(12)
    B, H, W, Cin, Cout = 2, 36, 44, 128, 64
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 4) ** 0.5
    masks = []
    for c in range(Cin // 16):
        y0, x0 = (c >> 1) & 1, c & 1
        m = 0
        for dy in range(2):
            for dx in range(2):
                m |= 1 << ((y0 + dy) * 3 + x0 + dx)
        masks.append(m)
    pc = K.pack_conv(w.cuda(), None)
    pc.tap_mask = torch.tensor(masks, dtype=torch.int32).cuda()
    src = K.to_cp16(_nhwc(x).cuda())                                                   # [B, 8, H, W, 16]
    hs = src.view(B, Cin // 16, H, W, 2, 8).permute(0, 1, 2, 4, 3, 5).contiguous().view(B, Cin // 16, H, W, 16)
    assert torch.equal(K.halfsplit_to_rows(hs), src)
    want = K.conv_ring(src, pc)
    got = K.conv_ring(hs, pc, src_halfsplit=True)
    torch.cuda.synchronize()
    assert torch.equal(got, want)


def test_conv3x3_ring_sparse_taps():
    """Four-taps-per-chunk form (the composed stride-2 convolution): weights zero outside a per-chunk 2x2 tap window."""
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(11)
    B, H, W, Cin, Cout = 2, 36, 44, 128, 64
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 4) ** 0.5
    masks = []
    for c in range(Cin // 16):
        y0, x0 = (c >> 1) & 1, c & 1                    # 2x2 window at (y0, x0)
        m = 0
        for dy in range(2):
            for dx in range(2):
                m |= 1 << ((y0 + dy) * 3 + x0 + dx)
        masks.append(m)
        keep = torch.zeros(3, 3)
        keep[y0:y0 + 2, x0:x0 + 2] = 1
        w[:, c * 16:(c + 1) * 16] *= keep
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.half().float(), w.half().float(), b, padding=1)
    pc = K.pack_conv(w.cuda(), b.cuda())
    pc.tap_mask = torch.tensor(masks, dtype=torch.int32, device="cuda")
    out = K.conv_ring(K.to_cp16(_nhwc(x).cuda()), pc)
    torch.cuda.synchronize()
    _cmp(out, ref, 1e-4, "ring conv, sparse taps")


@pytest.mark.parametrize("B,H,W,C", [(2, 9, 13, 64), (1, 16, 32, 32)])
def test_resample2_chunk_planar_fp16(B, H, W, C):
    """bilinear x2 written as the fp16 chunk-planar tensor vs F.interpolate (align_corners=False)."""
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(B + H + W)
    x = torch.randn(B, C, H, W, generator=g)
    ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    out = K.resample2(_nhwc(x).cuda(), up=True, cp16=True)
    torch.cuda.synchronize()
    assert tuple(out.shape) == (B, C // 16, 2 * H, 2 * W, 16)
    _cmp(K.from_cp16(out).float(), ref, 1e-3, "up2 -> cp16")


def test_conv3x3_ring_second_output_chunk_planar():
    """out2_cp16 of the ring kernel is the fp16 rounding of its fp32 result, in chunk-planar layout."""
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(3)
    B, H, W, Cin, Cout = 2, 20, 36, 64, 64
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / 24.0
    pc = K.pack_conv(w.cuda(), torch.randn(Cout, generator=g).cuda())
    res = torch.randn(B, H, W, Cout, generator=g).cuda()
    o16 = torch.zeros(B, Cout // 16, H, W, 16, dtype=torch.float16, device="cuda")
    out = K.conv_ring(K.to_cp16(_nhwc(x).cuda()), pc, res1=res, out2_cp16=o16)
    torch.cuda.synchronize()
    assert torch.equal(K.from_cp16(o16), out.half())


@pytest.mark.parametrize("B,H,W", [(2, 12, 30), (1, 16, 44), (2, 8, 64), (1, 8, 8), (3, 10, 202)])
def test_block_prologue(B, H, W):
    """u16 = bilinear_x2(up.0(x)), d16 = down.0(mean2x2(x)) vs the reference order of operations in torch-cpu
    (arch.py:378-406: up.0 / down.0 are applied AFTER the resampling there; they commute with it)."""
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(B + H + W)
    x = torch.randn(B, 64, H, W, generator=g)
    wu, bu = torch.randn(64, 64, 1, 1, generator=g) / 8.0, torch.randn(64, generator=g)
    wd, bd = torch.randn(64, 64, 1, 1, generator=g) / 8.0, torch.randn(64, generator=g)
    ref_u = F.conv2d(F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False), wu, bu)
    ref_d = F.conv2d(F.interpolate(x, scale_factor=0.5, mode="bilinear", align_corners=False), wd, bd)
    u16, d16 = K.block_prologue(_nhwc(x).cuda(), K.pack_block_prologue(wu.cuda(), bu.cuda(), wd.cuda(), bd.cuda()))
    torch.cuda.synchronize()
    _cmp(K.from_cp16(u16).float(), ref_u, 1.5e-3, "block prologue, x2 branch")
    _cmp(K.from_cp16(d16).float(), ref_d, 1.5e-3, "block prologue, x1/2 branch")


def test_conv3x3_ring_half_resolution_residual():
    """res_up2: the ring kernel adds bilinear_x2 of a half-resolution tensor in its epilogue."""
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(17)
    B, H, W, Cin, Cout = 2, 20, 36, 64, 64
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / 24.0
    b = torch.randn(Cout, generator=g)
    e = torch.randn(B, Cout, H // 2, W // 2, generator=g)
    ref = F.conv2d(x.half().float(), w.half().float(), b, padding=1) + \
        F.interpolate(e, scale_factor=2, mode="bilinear", align_corners=False)
    out = K.conv_ring(K.to_cp16(_nhwc(x).cuda()), K.pack_conv(w.cuda(), b.cuda()), res_up2=_nhwc(e).cuda())
    torch.cuda.synchronize()
    _cmp(out, ref, 1e-4, "ring conv + half-resolution residual")


@pytest.mark.parametrize("B,H,W", [(2, 36, 44), (1, 64, 96), (1, 56, 40), (1, 48, 70), (3, 16, 64), (2, 272, 480)])
def test_conv3x3_ring_sparse_taps_with_residuals(B, H, W):
    """Four-tap form with every epilogue input of Block_'s last convolution: res1, the half-resolution residual (staged
    in LDS by DMA in this form) and the fp16 chunk-planar second output; several tiles per workgroup at 272 x 480."""
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(B + H + W)
    Cin, Cout = 128, 64
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 4) ** 0.5
    masks = []
    for c in range(Cin // 16):
        y0, x0 = (c >> 1) & 1, c & 1
        masks.append(sum(1 << ((y0 + dy) * 3 + x0 + dx) for dy in range(2) for dx in range(2)))
        keep = torch.zeros(3, 3)
        keep[y0:y0 + 2, x0:x0 + 2] = 1
        w[:, c * 16:(c + 1) * 16] *= keep
    b = torch.randn(Cout, generator=g)
    r1 = torch.randn(B, Cout, H, W, generator=g)
    e = torch.randn(B, Cout, H // 2, W // 2, generator=g)
    ref = F.conv2d(x.half().float(), w.half().float(), b, padding=1) + r1 + \
        F.interpolate(e, scale_factor=2, mode="bilinear", align_corners=False)
    pc = K.pack_conv(w.cuda(), b.cuda())
    pc.tap_mask = torch.tensor(masks, dtype=torch.int32, device="cuda")
    o16 = torch.zeros(B, Cout // 16, H, W, 16, dtype=torch.float16, device="cuda")
    out = K.conv_ring(K.to_cp16(_nhwc(x).cuda()), pc, res1=_nhwc(r1).cuda(), res_up2=_nhwc(e).cuda(), out2_cp16=o16)
    torch.cuda.synchronize()
    _cmp(out, ref, 1e-4, "ring conv, four taps + residuals")
    assert torch.equal(K.from_cp16(o16), out.half())
    # hi | lo planes as the second output (a group's last block feeds the split-fp16 group convolution), and run-to-run equality
    ohl = torch.zeros(B, 2 * (Cout // 16), H, W, 16, dtype=torch.float16, device="cuda")
    out2 = K.conv_ring(K.to_cp16(_nhwc(x).cuda()), pc, res1=_nhwc(r1).cuda(), res_up2=_nhwc(e).cuda(), out2_cp16=ohl, out2_hl=True)
    torch.cuda.synchronize()
    assert torch.equal(out2, out)
    hl = K.from_cp16(ohl).float()
    assert torch.equal(hl[..., :Cout].half(), out.half())
    assert ((hl[..., :Cout] + hl[..., Cout:]) - out).abs().max().item() <= 2e-3 * 2.0 ** -10 * max(1.0, out.abs().max().item())


@pytest.mark.parametrize("B,H,W", [(2, 8, 24), (1, 12, 40)])
def test_conv_last(B, H, W):
    """conv_last (3x3, 64 -> 1) + bilinear x4 of the centre LR frame (arch.py:4476-4480)."""
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(B + H + W)
    feat = torch.randn(B, 64, 4 * H, 4 * W, generator=g)
    w = torch.randn(1, 64, 3, 3, generator=g) / 24.0
    b = torch.randn(1, generator=g)
    xc = torch.rand(B, 1, H, W, generator=g)
    ref = F.conv2d(feat, w, b, padding=1) + F.interpolate(xc, scale_factor=4, mode="bilinear", align_corners=False)
    out = K.conv_last(_nhwc(feat).cuda(), w.cuda().contiguous(), b.cuda(), xc.cuda().contiguous(), H * W)
    torch.cuda.synchronize()
    err = (out.cpu() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("B,H,W,two", [(2, 16, 40, False), (1, 24, 70, True), (2, 272, 480, True)])
def test_conv3x3_ws_residual_form(B, H, W, two):
    """conv3x3_ws_res: fp32 pixel-major act(conv + bias) + res1 (+ res2) and the fp16 chunk-planar copy."""
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(B + H + W)
    x = torch.randn(B, 64, H, W, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) / 24.0
    b = torch.randn(64, generator=g)
    r1, r2 = torch.randn(B, 64, H, W, generator=g), torch.randn(B, 64, H, W, generator=g)
    ref = F.conv2d(x.half().float(), w.half().float(), b, padding=1) + r1 + (r2 if two else 0)
    o16 = torch.zeros(B, 4, H, W, 16, dtype=torch.float16, device="cuda")
    out = K.conv3x3_ws_res(K.to_cp16(_nhwc(x).cuda()), K.pack_conv(w.cuda(), b.cuda()), res1=_nhwc(r1).cuda(),
                           res2=_nhwc(r2).cuda() if two else None, out2_cp16=o16)
    torch.cuda.synchronize()
    _cmp(out, ref, 1e-4, "ws conv, residual form")
    assert torch.equal(K.from_cp16(o16), out.half())


@pytest.mark.parametrize("B,H,W", [(2, 8, 24), (1, 12, 44)])
def test_upconv_last_fused_tail(B, H, W):
    """upconv2 + PixelShuffle + LeakyReLU + conv_last + bilinear x4 skip, without the HR feature map."""
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(B + H + W)
    x2 = torch.randn(B, 64, 2 * H, 2 * W, generator=g)          # features at 2x resolution
    wu, bu = torch.randn(256, 64, 1, 1, generator=g) / 8.0, torch.randn(256, generator=g) * 0.1
    wl, bl = torch.randn(1, 64, 3, 3, generator=g) / 24.0, torch.randn(1, generator=g)
    xc = torch.rand(B, 1, H, W, generator=g)
    hr = F.leaky_relu(F.pixel_shuffle(F.conv2d(x2, wu, bu), 2), 0.1)
    ref = F.conv2d(hr, wl, bl, padding=1) + F.interpolate(xc, scale_factor=4, mode="bilinear", align_corners=False)
    pc = K.pack_conv(wu.cuda(), bu.cuda(), shuffle2=True)
    out = K.upconv_last(_nhwc(x2).cuda(), pc, wl.cuda(), bl.cuda(), xc.cuda().contiguous(), H * W)
    torch.cuda.synchronize()
    err = (out.cpu() - ref).abs().max().item()
    assert err < 3e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("B,H,W", [(2, 20, 36), (1, 48, 64)])
def test_split_fp16_conv_on_ring_kernel(B, H, W):
    """LayerNorm -> fp16 hi | lo planes -> K-expanded (w_hi | w_hi | w_lo) 3x3 conv on the ring kernel: fp32-grade."""
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(B + H + W)
    x = torch.randn(B, 64, H, W, generator=g) * 3 + 0.5
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    w = torch.randn(64, 64, 3, 3, generator=g) / 24.0
    b = torch.randn(64, generator=g)
    r1, r2 = torch.randn(B, 64, H, W, generator=g), torch.randn(B, 64, H, W, generator=g)
    ln = F.layer_norm(x.permute(0, 2, 3, 1), (64,), gamma, beta, 1e-5).permute(0, 3, 1, 2)
    ref = F.conv2d(ln.double(), w.double(), b.double(), padding=1).float() + r1 + r2
    hl = K.layernorm64_hl(_nhwc(x).cuda(), gamma.cuda(), beta.cuda())
    out = K.conv_ring(hl, K.pack_conv_hilo(w.cuda(), b.cuda()), res1=_nhwc(r1).cuda(), res2=_nhwc(r2).cuda(), plane_wrap=8)
    torch.cuda.synchronize()
    _cmp(out, ref, 2e-5, "split-fp16 conv on the ring kernel")


@pytest.mark.parametrize("srcs,Cout,nres,per_image", [([64], 64, 2, True), ([64, 64], 64, 1, False), ([64, 128], 64, 0, True),
                                                      ([64], 128, 0, False), ([128], 128, 1, False)])
def test_conv1x1_stream_many_tiles_across_images(srcs, Cout, nres, per_image):
    """The persistent streaming form of the 1x1 convolution (conv1x1_stream.hip) with more tiles than workgroups, so
    that workgroups walk several tiles, cross image boundaries (weights reloaded per image) and end on a ragged tile;
    against torch, and against the one-tile-per-workgroup kernel it replaces (CDFO_CONV1X1_STREAM=0 is a process-wide
    switch, so that comparison is numerical: same split-bf16 arithmetic, fp32 accumulation)."""
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(sum(srcs) * 7 + Cout + nres)
    B, H, W = 5, 104, 105                     # 10,920 pixels = 85.3 tiles per image, 430 tiles in all
    xs = [torch.randn(B, c, H, W, generator=g) for c in srcs]
    cin = sum(srcs)
    b = torch.randn(Cout, generator=g)
    if per_image:
        ws = torch.randn(B, Cout, cin, 1, 1, generator=g) / cin ** 0.5
        y = torch.cat([F.conv2d(torch.cat([t[i:i + 1] for t in xs], 1), ws[i]) for i in range(B)], 0)
        packs = [K.pack_conv(ws[i].cuda(), None) for i in range(B)]
        pc = K.PackedConv(torch.stack([p.w for p in packs]).contiguous(), None, Cout, cin, 1, packs[0].CoutP, False,
                          packs[0].w.numel())
    else:
        w = torch.randn(Cout, cin, 1, 1, generator=g) / cin ** 0.5
        y = F.conv2d(torch.cat(xs, 1), w, b)
        pc = K.pack_conv(w.cuda(), b.cuda())
    ref = F.relu(y)
    rs = [torch.randn(B, Cout, H, W, generator=g) for _ in range(nres)]
    for r in rs:
        ref = ref + r
    dev_r = [_nhwc(r).cuda() for r in rs] + [None, None]
    out = K.conv([_nhwc(t).cuda() for t in xs], pc, act=K.ACT_RELU, res1=dev_r[0], res2=dev_r[1], prec=K.PREC_BF16X3)
    torch.cuda.synchronize()
    _cmp(out, ref, 3e-5, "conv1x1 stream")
    out2 = K.conv([_nhwc(t).cuda() for t in xs], pc, act=K.ACT_RELU, res1=dev_r[0], res2=dev_r[1], prec=K.PREC_BF16X3)
    torch.cuda.synchronize()
    assert torch.equal(out, out2)             # no atomics, fixed summation order


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W", [(2, 8, 8), (3, 17, 29), (1, 64, 96)])
def test_udsa_head_is_body0_of_conv_second(B, H, W):
    """cdfo_udsa_head composes conv_second (1 -> 64) with the prior U-net's body.0 (64 -> 16, LeakyReLU 0.1), borders
    included (arch.py:4420, 1819-1820): compared with the two torch convolutions in float64."""
    import torch.nn.functional as F
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(B * 100 + H)
    img = torch.randn(B, 1, H, W, generator=g)
    w2, b2 = torch.randn(64, 1, 3, 3, generator=g) / 3, torch.randn(64, generator=g) * 0.3
    w0, b0 = torch.randn(16, 64, 3, 3, generator=g) / 24, torch.randn(16, generator=g) * 0.3
    ref = F.leaky_relu(F.conv2d(F.conv2d(img.double(), w2.double(), b2.double(), padding=1), w0.double(), b0.double(), padding=1), 0.1)
    packed = K.pack_udsa_head(w0.cuda(), b0.cuda(), w2.cuda(), b2.cuda())
    out = K.udsa_head(img.cuda().contiguous(), H * W, B, H, W, packed)
    err = (out.permute(0, 3, 1, 2).double().cpu() - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.gpu
def test_stem_conv2_is_two_stencils_of_one_plane():
    """cdfo_stem_conv2: outA = conv(img; wA, bA) + add, outB = relu(conv(img; wB, bB)) (arch.py:4446-4449, 2200)."""
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(11)
    B, H, W = 3, 17, 29
    img = torch.randn(B, 1, H, W, generator=g)
    wA, bA = torch.randn(64, 1, 3, 3, generator=g) / 3, torch.randn(64, generator=g)
    wB, bB = torch.randn(64, 1, 3, 3, generator=g) / 3, torch.randn(64, generator=g)
    add = torch.randn(B, H, W, 64, generator=g)
    outA, outB = torch.empty(B, H, W, 64, device="cuda"), torch.empty(B, H, W, 64, device="cuda")
    K.stem_conv2(img.cuda().contiguous(), H * W, B, H, W, wA.cuda(), bA.cuda(), add.cuda(), outA, wB.cuda(), bB.cuda(), K.ACT_RELU, outB)
    refA = F.conv2d(img, wA, bA, padding=1).permute(0, 2, 3, 1) + add
    refB = F.relu(F.conv2d(img, wB, bB, padding=1)).permute(0, 2, 3, 1)
    assert (outA.cpu() - refA).abs().max().item() <= 1e-5
    assert (outB.cpu() - refB).abs().max().item() <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W", [(2, 16, 24), (3, 9, 50)])
def test_conv1x1_with_layernorm_planes_of_the_result(B, H, W):
    """conv(..., ln_out=(gamma, beta)): the 1x1 result plus LayerNorm64 of it as fp16 hi | lo planes (streaming kernel's
    epilogue), against the two separate kernels."""
    from cdfo_amd import kernels as K
    g = torch.Generator(device="cuda").manual_seed(B * 10 + H)
    x = torch.randn(B, H, W, 64, device="cuda", generator=g)
    res = torch.randn(B, H, W, 64, device="cuda", generator=g)
    gamma, beta = torch.rand(64, device="cuda", generator=g) + 0.5, torch.randn(64, device="cuda", generator=g) * 0.2
    pc = K.pack_conv(torch.randn(64, 64, 1, 1, device="cuda", generator=g) / 8, torch.randn(64, device="cuda", generator=g))
    out, planes = K.conv([x], pc, res1=res, prec=K.PREC_BF16X3, ln_out=(gamma, beta))
    ref = K.conv([x], pc, res1=res, prec=K.PREC_BF16X3)
    assert torch.equal(out, ref)
    ref_planes = K.layernorm64_hl(ref, gamma, beta)
    hl = planes.float()[:, 0:4] + planes.float()[:, 4:8]
    ref_hl = ref_planes.float()[:, 0:4] + ref_planes.float()[:, 4:8]
    assert (hl - ref_hl).abs().max().item() <= 2e-5
    assert (planes[:, 0:4].float() - ref_planes[:, 0:4].float()).abs().max().item() <= 4e-3      # hi halves: within an fp16 ulp


@pytest.mark.parametrize("B,H,W", [(2, 16, 32), (1, 13, 45), (3, 8, 8), (1, 272, 480)])
@pytest.mark.parametrize("act", [0, 1])
def test_conv3x3_n16_matches_fp64_conv(B, H, W, act):
    """cdfo_conv3x3_c64_n16 (the prior U-net's first layer, arch.py:1815-1834: Conv2d(64, 16, 3, 1, 1) [+ LeakyReLU]) against a float64
    convolution: split-bf16 three-pass arithmetic = fp32-grade; ragged tiles, a channel slice of a wider tensor as the source."""
    import torch.nn.functional as F
    from cdfo_amd import kernels as K
    g = torch.Generator(device="cuda").manual_seed(B * 100 + H + W)
    wide = torch.randn(B, H, W, 128, device="cuda", generator=g)
    x = wide[..., 32:96]                                            # pitch 128, 64 channels
    w = torch.randn(16, 64, 3, 3, device="cuda", generator=g) / 24
    b = torch.randn(16, device="cuda", generator=g)
    out = K.conv3x3_n16(x, K.pack_conv_n16(w), b, K.ACT_LRELU if act else K.ACT_NONE)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1)
    if act:
        ref = F.leaky_relu(ref, 0.1)
    torch.cuda.synchronize()
    assert tuple(out.shape) == (B, H, W, 16)
    err = (out.permute(0, 3, 1, 2).double() - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err


def test_conv1x1_with_its_fp16_chunk_planar_copy():
    """cp16_out: the streaming 1x1 kernel's second output == to_cp16 of its fp32 result (per-image weights, a ragged pixel count)."""
    from cdfo_amd import kernels as K
    g = torch.Generator(device="cuda").manual_seed(11)
    B, H, W = 3, 19, 45
    x = torch.randn(B, H, W, 64, device="cuda", generator=g)
    wt = torch.randn(B, 64, 64, 1, 1, device="cuda", generator=g) / 8
    pcs = [K.pack_conv(wt[i], None) for i in range(B)]
    pc = K.PackedConv(torch.stack([p.w for p in pcs]).view(B, -1).contiguous(), None, 64, 64, 1, 64, False, 4096)
    out, copy = K.conv(x, pc, prec=K.PREC_BF16X3, cp16_out=True)
    ref = K.conv(x, pc, prec=K.PREC_BF16X3)
    assert torch.equal(out, ref)
    assert tuple(copy.shape) == (B, 4, H, W, 16) and torch.equal(copy, K.to_cp16(ref))


@pytest.mark.parametrize("Cout,h,w,B,act", [(256, 12, 20, 1, 1), (128, 8, 32, 2, 0), (256, 18, 26, 2, 1), (256, 2, 2, 1, 1),
                                            (256, 136, 240, 2, 1), (256, 24, 336, 13, 1)])
def test_conv3x3_c64_wino_up2(Cout, h, w, B, act):
    """Block_'s x2 branch with the bilinear x2 folded into the Winograd input transform (cdfo_conv3x3_c64_wino_up2) vs torch-cpu
    float64: conv2d(F.interpolate(src, scale 2, bilinear, align_corners=False), padding=1) on the fp16-rounded low-resolution source --
    the arch.py:398-404 operator chain; clamped interpolation taps, zero padding of the x2 image, every edge tile, several units per
    workgroup.  Also against the materialised form (block_prologue's x2 image through conv3x3_wino)."""
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(Cout + h + w)
    x = torch.randn(B, 64, h, w, generator=g)
    wt = torch.randn(Cout, 64, 3, 3, generator=g) / 24.0
    b = torch.randn(Cout, generator=g)
    up = F.interpolate(x.half().double(), scale_factor=2.0, mode="bilinear", align_corners=False)
    ref = F.conv2d(up, wt.double(), b.double(), padding=1)
    ref = {0: ref, 1: F.leaky_relu(ref, 0.1)}[act].float()
    pc = K.pack_conv(wt.cuda(), b.cuda())
    src = K.to_cp16(_nhwc(x).cuda())
    out = K.conv3x3_wino_up2(src, pc, act=act)
    out_hs = K.halfsplit_to_rows(K.conv3x3_wino_up2(src, pc, act=act, halfsplit=True))      # CDFO_STORE_S2D_HS: the same values, rows half-split
    torch.cuda.synchronize()
    assert torch.equal(out_hs, out)
    got = K.from_cp16(out).float().cpu().view(B, h, w, 2, 2, Cout).permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * h, 2 * w, Cout).permute(0, 3, 1, 2)
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= 2.5e-3 * max(1.0, scale), f"wino up2: max-abs {err} (ref scale {scale})"


@pytest.mark.parametrize("B,H,W,stride,out_pad,hl", [
    (2, 6, 6, 2, 0, False), (3, 9, 70, 2, 1, False), (2, 35, 61, 2, 0, True), (1, 70, 122, 2, 0, False), (2, 137, 241, 2, 1, True),
    (2, 5, 130, 2, 1, False), (2, 7, 9, 1, 0, False)])
def test_small_conv16_transposed_four_pixels_per_thread(B, H, W, stride, out_pad, hl):
    """The prior U-net's transposed 16 -> 16 convolutions (arch/SIDECVSR_our.py:1827-1830) on the four-pixels-per-thread kernel
    (phase-major rows, odd and even output sizes, rows longer and shorter than a wave's 64 cells, the fp16 hi | lo plane output,
    and the stride-1 form the backward of a plain convolution uses) against torch conv_transpose2d in float64."""
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + W)
    x = torch.randn(B, 16, H, W, generator=g)
    w = torch.randn(16, 16, 3, 3, generator=g) / 6
    b = torch.randn(16, generator=g)
    pad = 2 if stride == 2 else 1
    ref = F.leaky_relu(F.conv_transpose2d(x.double(), w.double(), b.double(), stride=stride, padding=pad, output_padding=out_pad), 0.1).float()
    xg = _nhwc(x).cuda()
    got = K.small_conv16(xg, w.cuda(), b.cuda(), stride, pad, out_pad, True, K.ACT_LRELU, out_hl=hl)
    torch.cuda.synchronize()
    Ho, Wo = ref.shape[2], ref.shape[3]
    if hl:      # [B][2][Ho*Wo][16] fp16 hi | lo planes
        assert got.dtype == torch.float16
        pl = got.float().cpu().reshape(B, 2, Ho, Wo, 16)
        val = (pl[:, 0] + pl[:, 1]).permute(0, 3, 1, 2)
        tol = 5e-6 * max(1.0, ref.abs().max().item()) + 1e-6       # hi + lo carries ~22 bits
    else:
        val = got.cpu().permute(0, 3, 1, 2)
        tol = 5e-6 * max(1.0, ref.abs().max().item())
    assert tuple(val.shape) == tuple(ref.shape)
    err = (val - ref).abs().max().item()
    assert err <= tol, f"transposed small conv: {err} > {tol}"
