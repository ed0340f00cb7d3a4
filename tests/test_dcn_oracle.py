"""Pins the C oracle of the DCN forward (oracle/dcn_ref.c): the reference's own known-answer test
(ops/dcn/simple_check.py:8-22) plus derived identities against torch-cpu F.conv2d."""
import numpy as np
import torch
import torch.nn.functional as F

from dcn_oracle import dcn_forward_ref


def test_reference_simple_check_known_answer():
    # DeformConv(2, 1, 3, padding=1, deformable_groups=2), weights == 1, input arange(18).view(1,2,3,3); the offsets make
    # tap (i,j) sample the centre pixel, so every output = 9 * (x0 + x1).  Expected vector from simple_check.py:19.
    off = np.array([1, 1, 1, 0, 1, -1, 0, 1, 0, 0, 0, -1, -1, 1, -1, 0, -1, -1], np.float32)
    offset = np.tile(off.reshape(1, 18, 1, 1), (1, 2, 3, 3))
    x = np.arange(18, dtype=np.float32).reshape(1, 2, 3, 3)
    w = np.ones((1, 2, 3, 3), np.float32)
    out = dcn_forward_ref(x, offset, None, w, None, pad=1, dg=2)
    gt = np.array([81, 99, 117, 135, 153, 171, 189, 207, 225], np.float32)
    assert np.abs(gt - out.reshape(-1)).sum() < 1e-8


def _rand(shape, seed):
    return np.random.RandomState(seed).standard_normal(shape).astype(np.float32)


def test_zero_offset_unit_mask_is_conv2d():
    for (C, Co, k, s, p, d, g, dg) in [(8, 6, 3, 1, 1, 1, 1, 4), (8, 8, 3, 2, 1, 1, 2, 2), (4, 4, 3, 1, 2, 2, 1, 1),
                                       (6, 3, 1, 1, 0, 1, 3, 3)]:
        x, w, b = _rand((2, C, 9, 11), 1), _rand((Co, C // g, k, k), 2), _rand((Co,), 3)
        ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), s, p, d, g).numpy()
        Ho, Wo = ref.shape[2:]
        off = np.zeros((2, 2 * dg * k * k, Ho, Wo), np.float32)
        msk = np.ones((2, dg * k * k, Ho, Wo), np.float32)
        out = dcn_forward_ref(x, off, msk, w, b, s, p, d, g, dg)
        assert np.abs(out - ref).max() < 1e-5
        out1 = dcn_forward_ref(x, off, None, w, b, s, p, d, g, dg)      # mask == 1  <=>  DCNv1
        assert np.array_equal(out, out1)


def test_integer_offsets_are_a_shifted_conv_and_far_offsets_give_zero():
    x, w = _rand((1, 4, 10, 12), 5), _rand((5, 4, 3, 3), 6)
    off = np.zeros((1, 18, 10, 12), np.float32)
    off[:, 0::2] = 2.0   # every tap 2 rows down
    off[:, 1::2] = -1.0  # and 1 column left
    out = dcn_forward_ref(x, off, None, w, None, pad=1)
    xs = np.zeros_like(x)
    xs[:, :, :-2, 1:] = x[:, :, 2:, :-1]          # xs[y][x] = x[y+2][x-1]
    ref = F.conv2d(torch.from_numpy(xs), torch.from_numpy(w), None, 1, 1).numpy()
    assert np.abs(out[:, :, 1:-1, 1:-1] - ref[:, :, 1:-1, 1:-1]).max() < 1e-5
    off[:] = 1000.0
    assert np.abs(dcn_forward_ref(x, off, None, w, None, pad=1)).max() == 0.0


def test_mask_scales_linearly_and_fractional_offsets_interpolate():
    x, w = _rand((1, 2, 6, 6), 7), _rand((3, 2, 3, 3), 8)
    off = np.full((1, 18, 6, 6), 0.5, np.float32)
    m1 = np.random.RandomState(9).uniform(0, 1, (1, 9, 6, 6)).astype(np.float32)
    a = dcn_forward_ref(x, off, m1, w, None, pad=1)
    b = dcn_forward_ref(x, off, 2 * m1, w, None, pad=1)
    assert np.abs(2 * a - b).max() < 1e-5
    # half-pixel offsets == conv over the 2x2-averaged image (interior only; borders differ by the zero rule)
    xa = 0.25 * (x[:, :, :-1, :-1] + x[:, :, 1:, :-1] + x[:, :, :-1, 1:] + x[:, :, 1:, 1:])
    ref = F.conv2d(torch.from_numpy(xa), torch.from_numpy(w), None).numpy()       # valid conv on the 5x5 average
    full = dcn_forward_ref(x, off, None, w, None, pad=1)
    assert np.abs(full[:, :, 1:4, 1:4] - ref).max() < 1e-5


# ------------------------------------------------------------------------------------------------ backward restatement
def _dcn_forward_f64(x, offset, mask, w, b, s, p, d, groups, dg):
    """Independent, differentiable statement of the operator (float64, gather based) used only to pin the C oracle's
    backward: out = sum_taps W * mask * bilinear(x, base + offset), corners outside the image contribute 0."""
    B, C, H, W = x.shape
    Co, Cg, kh, kw = w.shape
    T = kh * kw
    Ho = (H + 2 * p - (d * (kh - 1) + 1)) // s + 1
    Wo = (W + 2 * p - (d * (kw - 1) + 1)) // s + 1
    Cdg = C // dg
    ys = (torch.arange(Ho, dtype=x.dtype) * s - p).view(1, 1, Ho, 1)
    xs = (torch.arange(Wo, dtype=x.dtype) * s - p).view(1, 1, 1, Wo)
    cols = []
    for t in range(T):
        i, j = divmod(t, kw)
        off = offset.view(B, dg, T, 2, Ho, Wo)[:, :, t]
        hy = ys + i * d + off[:, :, 0]                                       # [B, dg, Ho, Wo]
        wx = xs + j * d + off[:, :, 1]
        hl, wl = torch.floor(hy).detach(), torch.floor(wx).detach()
        lh, lw = hy - hl, wx - wl
        xg = x.view(B, dg, Cdg, H * W)
        val = 0
        for (dy, dx, wt) in ((0, 0, (1 - lh) * (1 - lw)), (0, 1, (1 - lh) * lw), (1, 0, lh * (1 - lw)), (1, 1, lh * lw)):
            yy, xx = hl + dy, wl + dx
            ok = ((yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1)).to(x.dtype)
            idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).long().view(B, dg, 1, Ho * Wo).expand(B, dg, Cdg, Ho * Wo)
            val = val + torch.gather(xg, 3, idx).view(B, dg, Cdg, Ho, Wo) * (wt * ok).unsqueeze(2)
        if mask is not None:
            val = val * mask.view(B, dg, T, Ho, Wo)[:, :, t].unsqueeze(2)
        cols.append(val.reshape(B, C, Ho, Wo))
    col = torch.stack(cols, 2)                                               # [B, C, T, Ho, Wo]
    col = col.view(B, groups, Cg * T, Ho * Wo)
    out = torch.einsum("gok,bgkp->bgop", w.view(groups, Co // groups, Cg * T), col).reshape(B, Co, Ho, Wo)
    return out if b is None else out + b.view(1, -1, 1, 1)


def test_backward_restatement_matches_float64_autograd():
    from oracle.dcn_modules_ref import dcn_backward_ref
    for n, (C, Co, k, s, p, d, g, dg, modulated) in enumerate([(8, 6, 3, 1, 1, 1, 1, 4, True), (8, 8, 3, 2, 1, 1, 2, 2, True),
                                                              (4, 4, 3, 1, 2, 2, 1, 1, False), (6, 3, 1, 1, 0, 1, 3, 3, True),
                                                              (4, 2, 3, 1, 1, 1, 2, 1, False)]):
        rs = np.random.RandomState(100 + n)
        B, H, W = 2, 7, 9
        Ho = (H + 2 * p - (d * (k - 1) + 1)) // s + 1
        Wo = (W + 2 * p - (d * (k - 1) + 1)) // s + 1
        x = rs.standard_normal((B, C, H, W)).astype(np.float32)
        w = rs.standard_normal((Co, C // g, k, k)).astype(np.float32)
        b = rs.standard_normal((Co,)).astype(np.float32)
        off = (3.0 * rs.standard_normal((B, 2 * dg * k * k, Ho, Wo))).astype(np.float32)     # many samples leave the image
        msk = rs.uniform(0, 1, (B, dg * k * k, Ho, Wo)).astype(np.float32) if modulated else None
        go = rs.standard_normal((B, Co, Ho, Wo)).astype(np.float32)
        got = dcn_backward_ref(x, off, msk, w, go, s, p, d, g, dg, scale=1.0, with_bias=True)
        tx, tw, tb, toff = [torch.from_numpy(a).double().requires_grad_() for a in (x, w, b, off)]
        tm = None if msk is None else torch.from_numpy(msk).double().requires_grad_()
        out = _dcn_forward_f64(tx, toff, tm, tw, tb, s, p, d, g, dg)
        fwd = dcn_forward_ref(x, off, msk, w, b, s, p, d, g, dg)
        assert np.abs(out.detach().numpy() - fwd).max() < 1e-4                 # the two forward statements agree
        out.backward(torch.from_numpy(go).double())
        want = dict(grad_input=tx.grad, grad_offset=toff.grad, grad_weight=tw.grad, grad_bias=tb.grad)
        if tm is not None:
            want["grad_mask"] = tm.grad
        for key, ref in want.items():
            ref = ref.numpy()
            err = np.abs(got[key] - ref).max() / max(1.0, np.abs(ref).max())
            assert err < 2e-5, (n, key, err)
        # `scale` multiplies the weight gradient only (cpp:373-378, deform_conv.py:92)
        half = dcn_backward_ref(x, off, msk, w, go, s, p, d, g, dg, scale=0.5, with_bias=True)
        assert np.allclose(half["grad_weight"], 0.5 * got["grad_weight"], atol=1e-5)
        assert np.array_equal(half["grad_input"], got["grad_input"])


def test_double_instantiation_agrees_with_float_and_with_float64_autograd():
    """oracle/dcn_ref_f64.c (the checker of the operator's fp64 entry points): same results as the float build to fp32
    rounding on the forward and the backward, and its gradients match float64 torch autograd through an independent
    gather-based statement of the forward (zero offsets + unit mask == conv2d)."""
    from oracle.dcn_modules_ref import dcn_backward_ref
    rs = np.random.RandomState(12)
    B, C, Co, H, W, k, dg = 2, 8, 6, 7, 9, 3, 4
    x, w, b = rs.standard_normal((B, C, H, W)), rs.standard_normal((Co, C, k, k)) / 8, rs.standard_normal((Co,))
    off = rs.standard_normal((B, 2 * dg * k * k, H, W)) * 2
    msk = rs.uniform(0, 1, (B, dg * k * k, H, W))
    go = rs.standard_normal((B, Co, H, W))
    f32 = dcn_forward_ref(x, off, msk, w, b, 1, 1, 1, 1, dg)
    f64 = dcn_forward_ref(x, off, msk, w, b, 1, 1, 1, 1, dg, dtype=np.float64)
    assert f64.dtype == np.float64 and np.abs(f64 - f32).max() < 2e-5
    g32 = dcn_backward_ref(x, off, msk, w, go, 1, 1, 1, 1, dg)
    g64 = dcn_backward_ref(x, off, msk, w, go, 1, 1, 1, 1, dg, dtype=np.float64)
    for key in g64:
        assert g64[key].dtype == np.float64
        assert np.abs(g64[key] - g32[key]).max() <= 3e-5 * max(1.0, np.abs(g64[key]).max()), key
    # float64 pin: zero offsets + unit mask is a plain convolution, whose gradients autograd gives exactly
    z, one = np.zeros_like(off), np.ones_like(msk)
    xt, wt, bt = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (x, w, b))
    y = F.conv2d(xt, wt, bt, 1, 1)
    y.backward(torch.tensor(go))
    g = dcn_backward_ref(x, z, one, w, go, 1, 1, 1, 1, dg, dtype=np.float64)
    assert np.abs(dcn_forward_ref(x, z, one, w, b, 1, 1, 1, 1, dg, dtype=np.float64) - y.detach().numpy()).max() < 1e-12
    assert np.abs(g["grad_input"] - xt.grad.numpy()).max() < 1e-12
    assert np.abs(g["grad_weight"] - wt.grad.numpy()).max() < 1e-11
    assert np.abs(g["grad_bias"] - bt.grad.numpy()).max() < 1e-11
