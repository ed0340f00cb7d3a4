"""Each autograd Function of cdfo_amd/autograd.py (HIP forward + HIP backward) against torch-cpu float64 autograd of the same
operator written with stock torch ops (the operator definitions the oracle uses): outputs and every gradient."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def check(fn_hip, fn_ref, inputs, tol=2e-4, nhwc_in=(), nhwc_out=True, name=""):
    """inputs: dict name -> cpu float32 tensor (NCHW for pixel tensors listed in nhwc_in).  fn_hip gets cuda tensors (pixel
    tensors pixel-major), fn_ref float64 cpu tensors (NCHW).  Compares output and the gradient of every input."""
    g = torch.Generator().manual_seed(99)
    ref_in = {k: v.double().clone().requires_grad_(True) for k, v in inputs.items()}
    out_ref = fn_ref(**ref_in)
    go = torch.randn(out_ref.shape, generator=g, dtype=torch.float64)
    out_ref.backward(go)
    hip_in = {k: (nhwc(v) if k in nhwc_in else v.clone()).cuda().requires_grad_(True) for k, v in inputs.items()}
    out = fn_hip(**hip_in)
    go_h = (nhwc(go) if nhwc_out else go).float().cuda()
    out.backward(go_h)
    torch.cuda.synchronize()
    o = nchw(out.detach().cpu()) if nhwc_out else out.detach().cpu()
    scale = max(1.0, out_ref.abs().max().item())
    assert (o.double() - out_ref.detach()).abs().max().item() <= tol * scale, f"{name}: forward"
    for k in inputs:
        gr = ref_in[k].grad
        gh = hip_in[k].grad
        assert gh is not None, f"{name}: no gradient for {k}"
        gh = gh.detach().cpu()
        if k in nhwc_in:
            gh = nchw(gh)
        s = max(1e-6, gr.abs().max().item())
        err = (gh.double() - gr).abs().max().item()
        print(f"   {name}: d/d{k} rel err {err / s:.2e}")
        assert err <= tol * s, f"{name}: d/d{k}: {err:.3e} vs scale {s:.3e}"


def R(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("cs,Co,ks,act,nres", [([64], 64, 3, 1, 1), ([64, 64], 64, 3, 0, 0), ([64], 256, 3, 1, 0), ([256], 64, 3, 0, 2),
                                               ([64], 192, 1, 0, 0), ([64] * 7, 64, 1, 1, 0), ([64], 16, 3, 1, 0), ([16], 64, 3, 1, 1),
                                               ([64, 64], 64, 1, 2, 0)])
def test_conv(cs, Co, ks, act, nres):
    from cdfo_amd import autograd as A
    B, H, W = 2, 10, 12
    inp = {f"x{i}": R(B, c, H, W, seed=i) for i, c in enumerate(cs)}
    inp["w"] = R(Co, sum(cs), ks, ks, seed=20, scale=1.0 / (sum(cs) * ks * ks) ** 0.5)
    inp["b"] = R(Co, seed=21)
    for i in range(nres):
        inp[f"r{i}"] = R(B, Co, H, W, seed=30 + i)
    pix = [k for k in inp if k[0] in "xr"]
    actf = {0: lambda t: t, 1: lambda t: F.leaky_relu(t, 0.1), 2: F.relu}[act]

    def ref(**t):
        y = actf(F.conv2d(torch.cat([t[f"x{i}"] for i in range(len(cs))], 1), t["w"], t["b"], padding=ks // 2))
        for i in range(nres):
            y = y + t[f"r{i}"]
        return y

    def hip(**t):
        return A.conv([t[f"x{i}"] for i in range(len(cs))], t["w"], t["b"], 1, ks // 2, act, res=[t[f"r{i}"] for i in range(nres)])

    check(hip, ref, inp, nhwc_in=pix, name=f"conv {cs}->{Co} k{ks}")


def test_stem_and_conv_last():
    from cdfo_amd import autograd as A
    B, H, W = 3, 9, 11
    img = R(B, H, W, seed=1)
    inp = {"w": R(64, 1, 3, 3, seed=2, scale=0.3), "b": R(64, seed=3)}
    check(lambda w, b: A.stem(img.cuda(), w, b, 1), lambda w, b: F.leaky_relu(F.conv2d(img.double().unsqueeze(1), w, b, padding=1), 0.1),
          inp, name="stem")
    xc = R(B, 1, H, W, seed=4)
    inp = {"t": R(B, 64, 4 * H, 4 * W, seed=5), "w": R(1, 64, 3, 3, seed=6, scale=0.05), "b": R(1, seed=7)}
    check(lambda t, w, b: A.conv_last(t, w, b, xc.cuda().contiguous(), H * W),
          lambda t, w, b: F.conv2d(t, w, b, padding=1) + F.interpolate(xc.double(), scale_factor=4.0, mode="bilinear", align_corners=False),
          inp, nhwc_in=["t"], nhwc_out=False, name="conv_last")


def test_layernorm_dwconv():
    from cdfo_amd import autograd as A

    def ln_ref(x, gamma, beta):
        mu = x.mean(1, keepdim=True)
        var = x.var(1, keepdim=True, unbiased=False)
        return (x - mu) / torch.sqrt(var + 1e-5) * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)

    check(lambda x, gamma, beta: A.layernorm(x, gamma, beta), ln_ref,
          {"x": R(2, 64, 9, 13, seed=1, scale=2.0) + 0.5, "gamma": R(64, seed=2), "beta": R(64, seed=3)}, nhwc_in=["x"], name="layernorm")
    check(lambda x, w: A.dwconv(x, w), lambda x, w: F.conv2d(x, w, None, padding=1, groups=192),
          {"x": R(2, 192, 9, 13, seed=4), "w": R(192, 1, 3, 3, seed=5, scale=0.3)}, nhwc_in=["x"], name="dwconv")


@pytest.mark.parametrize("transposed,out_pad,H,W", [(False, 0, 16, 16), (False, 0, 9, 17), (True, 0, 6, 6), (True, 1, 9, 9), (True, 1, 5, 10)])
def test_small_conv16(transposed, out_pad, H, W):
    from cdfo_amd import autograd as A
    inp = {"x": R(2, 16, H, W, seed=1), "w": R(16, 16, 3, 3, seed=2, scale=0.1), "b": R(16, seed=3)}
    if transposed:
        ref = lambda x, w, b: F.leaky_relu(F.conv_transpose2d(x, w, b, stride=2, padding=2, output_padding=out_pad), 0.1)   # noqa: E731
    else:
        ref = lambda x, w, b: F.leaky_relu(F.conv2d(x, w, b, stride=2, padding=2), 0.1)                                      # noqa: E731
    check(lambda x, w, b: A.small_conv16(x, w, b, 2, 2, out_pad, transposed, 1), ref, inp, nhwc_in=["x"], name="small_conv16")


def test_spatial_gate16():
    from cdfo_amd import autograd as A

    def ref(t, w, b):
        pooled = torch.cat([t.max(1, keepdim=True)[0], t.mean(1, keepdim=True)], 1)
        return t * torch.sigmoid(F.conv2d(pooled, w, b, padding=3))

    check(lambda t, w, b: A.spatial_gate16(t, w, b), ref,
          {"t": R(2, 16, 10, 12, seed=1), "w": R(1, 2, 7, 7, seed=2, scale=0.2), "b": R(1, seed=3)}, nhwc_in=["t"], name="spatial_gate16")


@pytest.mark.parametrize("heads", [8, 4])
def test_channel_attention(heads):
    from cdfo_amd import autograd as A
    from oracle.cvsr_v8_ref import _channel_attention
    inp = {"q": R(3, 64, 12, 10, seed=1), "k": R(3, 64, 12, 10, seed=2), "v": R(3, 64, 12, 10, seed=3),
           "temp": R(heads, 1, 1, seed=4).abs() + 0.5}
    check(lambda q, k, v, temp: A.channel_attention(q, k, v, temp, heads), lambda q, k, v, temp: _channel_attention(q, k, v, heads, temp),
          inp, nhwc_in=["q", "k", "v"], name=f"channel_attention h{heads}")


def test_gates_masks_means():
    from cdfo_amd import autograd as A
    x = {"x": R(3, 64, 8, 12, seed=1), "gate": torch.rand(3, 64, generator=torch.Generator().manual_seed(2))}
    check(lambda x, gate: A.scale_channels(x, gate), lambda x, gate: x * gate.view(3, 64, 1, 1), x, nhwc_in=["x"], name="scale_channels")
    check(lambda x: A.chan_mean(x), lambda x: x.mean((2, 3)), {"x": x["x"]}, nhwc_in=["x"], nhwc_out=False, name="chan_mean")
    mask = (torch.rand(3, 64, 8, 12, generator=torch.Generator().manual_seed(3)) > 0.5).float()
    for inv in (False, True):
        check(lambda x: A.mul_mask(x, nhwc(mask).cuda(), inv), lambda x: x * ((1 - mask.double()) if inv else mask.double()),
              {"x": x["x"]}, nhwc_in=["x"], name="mul_mask")
    check(lambda a, b: A.add(a, b), lambda a, b: a + b, {"a": x["x"], "b": R(3, 64, 8, 12, seed=5)}, nhwc_in=["a", "b"], name="add")


def test_nine_tap_convolutions():
    from cdfo_amd import autograd as A
    B, H, W = 2, 12, 10
    x = R(B, 64, H, W, seed=1)

    def chan_ref(x, w, b):      # [b,c,h,w] -> rows [(b h), w, c], kernel (1,9) slides over c (arch.py:2216-2219)
        z = x.permute(0, 2, 3, 1).reshape(B * H, 1, W, 64)
        y = F.conv2d(z, w, b, padding=(0, 4))
        return y.reshape(B, H, W, 64).permute(0, 3, 1, 2)

    def col_ref(x, w, b):       # [(b w), 1, h, c], kernel (9,1) slides over h (arch.py:2225)
        z = x.permute(0, 3, 2, 1).reshape(B * W, 1, H, 64)
        y = F.conv2d(z, w, b, padding=(4, 0))
        return y.reshape(B, W, H, 64).permute(0, 3, 2, 1)

    check(lambda x, w, b: A.chanconv9(x, w, b), chan_ref, {"x": x, "w": R(1, 1, 1, 9, seed=2, scale=0.4), "b": R(1, seed=3)}, nhwc_in=["x"],
          name="chanconv9")
    check(lambda x, w, b: A.colconv9(x, w, b), col_ref, {"x": x, "w": R(1, 1, 9, 1, seed=4, scale=0.4), "b": R(1, seed=5)}, nhwc_in=["x"],
          name="colconv9")


@pytest.mark.parametrize("mode,H,W", [(0, 8, 16), (1, 16, 8), (2, 16, 24), (0, 8, 72), (1, 80, 8)])
def test_seq_attn(mode, H, W):
    from cdfo_amd import autograd as A
    B = 2
    inp = {"q": R(B, 64, H, W, seed=1, scale=0.35), "v": R(B, 64, H, W, seed=2)}

    def ref(q, v):
        if mode == 0:
            f = lambda z: z.permute(0, 2, 3, 1).reshape(B * H, W, 64)                          # noqa: E731
            back = lambda o: o.reshape(B, H, W, 64).permute(0, 3, 1, 2)                         # noqa: E731
        elif mode == 1:
            f = lambda z: z.permute(0, 3, 2, 1).reshape(B * W, H, 64)                          # noqa: E731
            back = lambda o: o.reshape(B, W, H, 64).permute(0, 3, 2, 1)                         # noqa: E731
        else:
            f = lambda z: z.reshape(B, 64, H // 8, 8, W // 8, 8).permute(0, 2, 4, 3, 5, 1).reshape(-1, 64, 64)   # noqa: E731
            back = lambda o: o.reshape(B, H // 8, W // 8, 8, 8, 64).permute(0, 5, 1, 3, 2, 4).reshape(B, 64, H, W)  # noqa: E731
        s = f(q)
        return back((s @ s.transpose(-2, -1)).softmax(-1) @ f(v))

    check(lambda q, v: A.seq_attn(q, v, mode), ref, inp, nhwc_in=["q", "v"], tol=5e-4, name=f"seq_attn mode {mode}")


def test_flow_warp_and_resamples():
    from cdfo_amd import autograd as A
    from oracle.cvsr_v8_ref import flow_warp
    B, H, W = 2, 12, 16
    mv = R(B, 2, H, W, seed=3, scale=2.5)
    check(lambda x: A.flow_warp(x, mv.cuda().contiguous(), 2 * H * W), lambda x: flow_warp(x, mv.double().permute(0, 2, 3, 1)),
          {"x": R(B, 64, H, W, seed=1)}, nhwc_in=["x"], name="flow_warp")
    check(lambda x: A.resample2(x, True), lambda x: F.interpolate(x, scale_factor=2.0, mode="bilinear", align_corners=False),
          {"x": R(B, 64, 6, 10, seed=2)}, nhwc_in=["x"], name="resample x2")
    check(lambda x: A.resample2(x, False), lambda x: F.interpolate(x, scale_factor=0.5, mode="bilinear", align_corners=False),
          {"x": R(B, 64, 12, 16, seed=4)}, nhwc_in=["x"], name="resample x0.5")


def test_gumbel_mask_matches_the_oracle_and_the_inference_kernel():
    from cdfo_amd import autograd as A
    from oracle.cvsr_v8_ref import gumbel_hard_mask
    B, H, W = 2, 16, 24
    g = torch.Generator().manual_seed(5)
    vmax = torch.rand(B, 64, generator=g) * 3
    u = torch.rand(B, 64, H, W, generator=g).clamp_min(1e-6)
    m = A.gumbel_mask(vmax.cuda(), u.cuda(), B, H, W)
    want = gumbel_hard_mask(vmax.view(B, 64, 1, 1).expand(B, 64, H, W), u)
    assert torch.equal(nchw(m.cpu()), want)
    cap = []
    m2 = A.gumbel_mask(vmax.cuda(), ("rng", 1234567, 3), B, H, W, cap)       # drawn in the kernel; the captured draws reproduce it
    want2 = gumbel_hard_mask(vmax.view(B, 64, 1, 1).expand(B, 64, H, W), cap[0].cpu())
    assert torch.equal(nchw(m2.cpu()), want2) and 0.0 < cap[0].min().item() and cap[0].max().item() < 1.0
