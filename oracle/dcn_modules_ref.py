"""ORACLE (test infrastructure, not product code): torch-cpu restatements of the two DCN consumer modules,
``DSTA.forward`` (ops/attentionlayer.py:117-156) and ``MVDualAttAlignment.forward`` (arch/SIDECVSR_our.py:3303-3352), over
a ``state_dict``; the deformable convolution inside them is the C oracle ``oracle/dcn_ref.c``.

Pinned: ``oracle/gen_fixtures.py`` runs the REAL reference classes on seeded inputs with only their deformable-conv call
(the absent ``deform_conv_cuda`` extension / ``torchvision.ops.deform_conv2d``) replaced by that C oracle, and commits
the outputs under tests/golden/ (dsta_*.npz, mvalign_*.npz).  The DCN step itself has no reference-owned vector beyond
the DCNv1 known answer (parity unpinned at that boundary)."""
import ctypes as C
import os
import subprocess

import numpy as np
import torch
import torch.nn.functional as F

from .cvsr_v8_ref import _channel_attention, flow_warp

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "_build", "libdcn_ref.so")
_lib = None


def _dcn_lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.run(["make", "-s", "-C", os.path.join(_ROOT, "oracle")], check=True)
        _lib = C.CDLL(_SO)
        _lib.dcn_forward_ref.restype = C.c_int
        _lib.dcn_forward_ref.argtypes = [C.c_void_p] * 6 + [C.c_int] * 15
        _lib.dcn_backward_ref.restype = C.c_int
        _lib.dcn_backward_ref.argtypes = [C.c_void_p] * 10 + [C.c_int] * 15 + [C.c_float]
        _lib.dcn_forward_ref_f64.restype = C.c_int
        _lib.dcn_forward_ref_f64.argtypes = [C.c_void_p] * 6 + [C.c_int] * 15
        _lib.dcn_backward_ref_f64.restype = C.c_int
        _lib.dcn_backward_ref_f64.argtypes = [C.c_void_p] * 10 + [C.c_int] * 15 + [C.c_double]
    return _lib


def dcn_forward_ref(x, offset, mask, weight, bias, stride=1, pad=0, dil=1, groups=1, dg=1, dtype=np.float32):
    """numpy arrays in the reference's layouts -> output [B,Co,Ho,Wo] (DCNv1 when mask is None); dtype float32 (default)
    or float64 (the C oracle's double instantiation)."""
    x = np.ascontiguousarray(x, dtype)
    offset = np.ascontiguousarray(offset, dtype)
    weight = np.ascontiguousarray(weight, dtype)
    mask = None if mask is None else np.ascontiguousarray(mask, dtype)
    bias = None if bias is None else np.ascontiguousarray(bias, dtype)
    B, Cc, H, W = x.shape
    Co, _, kh, kw = weight.shape
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    dh, dw = (dil, dil) if isinstance(dil, int) else dil
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    out = np.empty((B, Co, Ho, Wo), dtype)
    p = lambda a: None if a is None else a.ctypes.data  # noqa: E731
    fn = _dcn_lib().dcn_forward_ref if dtype == np.float32 else _dcn_lib().dcn_forward_ref_f64
    rc = fn(p(x), p(offset), p(mask), p(weight), p(bias), p(out), B, Cc, H, W, Co, kh, kw, sh, sw,
                                    ph, pw, dh, dw, groups, dg)
    if rc != 0:
        raise ValueError("dcn_forward_ref rejected the shapes")
    return out


def dcn_backward_ref(x, offset, mask, weight, gout, stride=1, pad=0, dil=1, groups=1, dg=1, scale=1.0, with_bias=True,
                     dtype=np.float32):
    """numpy fp32 arrays -> dict(grad_input, grad_offset, grad_mask (DCNv2 only), grad_weight, grad_bias) computed by
    oracle/dcn_ref.c:dcn_backward_ref from zero-initialised gradients, as the reference's autograd Functions do
    (ops/dcn/deform_conv.py:60-99, 150-172)."""
    f = lambda a: None if a is None else np.ascontiguousarray(a, dtype)  # noqa: E731
    x, offset, mask, weight, gout = f(x), f(offset), f(mask), f(weight), f(gout)
    B, Cc, H, W = x.shape
    Co, _, kh, kw = weight.shape
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    dh, dw = (dil, dil) if isinstance(dil, int) else dil
    g = dict(grad_input=np.zeros_like(x), grad_offset=np.zeros_like(offset), grad_weight=np.zeros_like(weight))
    if mask is not None:
        g["grad_mask"] = np.zeros_like(mask)
    if with_bias:
        g["grad_bias"] = np.zeros((Co,), dtype)
    p = lambda a: None if a is None else a.ctypes.data  # noqa: E731
    fn = _dcn_lib().dcn_backward_ref if dtype == np.float32 else _dcn_lib().dcn_backward_ref_f64
    rc = fn(p(x), p(offset), p(mask), p(weight), p(gout), p(g["grad_input"]), p(g["grad_offset"]),
                                     p(g.get("grad_mask")), p(g["grad_weight"]), p(g.get("grad_bias")), B, Cc, H, W, Co,
                                     kh, kw, sh, sw, ph, pw, dh, dw, groups, dg, float(scale))
    if rc != 0:
        raise ValueError("dcn_backward_ref rejected the shapes")
    return g


class _DcnRefFn(torch.autograd.Function):
    """The C oracle's forward AND backward (oracle/dcn_ref.c: dcn_forward_ref / dcn_backward_ref, float or double build) as a
    torch autograd node, so that torch autograd through the restated consumer modules is the gradient oracle of the HIP
    training paths (tests/test_gpu_dcn_modules.py)."""

    @staticmethod
    def forward(ctx, x, offset, mask, weight, bias, stride, pad, dil, groups, dg):
        dt = np.float64 if x.dtype == torch.float64 else np.float32
        t = lambda a: None if a is None else a.detach().numpy()  # noqa: E731
        ctx.meta = (stride, pad, dil, groups, dg, dt, bias is not None)
        ctx.save_for_backward(x, offset, mask, weight)
        return torch.from_numpy(dcn_forward_ref(t(x), t(offset), t(mask), t(weight), t(bias), stride, pad, dil, groups, dg, dtype=dt))

    @staticmethod
    def backward(ctx, g):
        x, offset, mask, weight = ctx.saved_tensors
        stride, pad, dil, groups, dg, dt, has_bias = ctx.meta
        t = lambda a: None if a is None else a.detach().numpy()  # noqa: E731
        r = dcn_backward_ref(t(x), t(offset), t(mask), t(weight), g.detach().contiguous().numpy(), stride, pad, dil, groups, dg,
                             with_bias=has_bias, dtype=dt)
        f = lambda k: torch.from_numpy(r[k]) if k in r else None  # noqa: E731
        return f("grad_input"), f("grad_offset"), f("grad_mask"), f("grad_weight"), (f("grad_bias") if has_bias else None), None, None, None, None, None


def dcn_torch(x, offset, mask, weight, bias, stride=1, pad=0, dil=1, groups=1, dg=1):
    """The C oracle on torch tensors (float32 or float64 by x.dtype); differentiable (see _DcnRefFn)."""
    c = lambda a: None if a is None else a.contiguous()  # noqa: E731
    return _DcnRefFn.apply(c(x), c(offset), c(mask), c(weight), c(bias), stride, pad, dil, groups, dg)


def dsta_forward(sd, x):
    f = sd["conv1.weight"].shape[0]
    cv = lambda k, t, s=1, p=0: F.conv2d(t, sd[k + ".weight"], sd[k + ".bias"], stride=s, padding=p)  # noqa: E731
    c1_ = cv("conv1", x)
    c1 = cv("conv2", c1_, 2, 0)
    v_max = F.max_pool2d(c1, kernel_size=7, stride=3)
    v_range = F.relu(cv("conv_max", v_max, 1, 1))
    c3 = F.relu(cv("conv3", v_range, 1, 1))
    c3 = F.relu(cv("conv3_", c3, 1, 1))
    dc3 = F.relu(cv("down_conv2.0", c3, 2, 1))
    off_mask2 = cv("mask2", dc3, 1, 1)
    off_msk = cv("mask", c3, 1, 1)
    off_msk = off_msk + F.interpolate(off_mask2, off_msk.shape[2:], mode="bilinear", align_corners=False)
    off, msk = off_msk[:, :f * 18], torch.sigmoid(off_msk[:, f * 18:])
    c3 = F.relu(dcn_torch(v_max, off, msk, sd["dcn.weight"], sd["dcn.bias"], 1, 1, 1, 1, f))
    y = c3.mean((2, 3), keepdim=True)
    y = torch.sigmoid(cv("conv_du.2", F.relu(cv("conv_du.0", y))))
    c3 = F.interpolate(c3, x.shape[2:], mode="bilinear", align_corners=False)
    c4 = cv("conv4", c3 + cv("conv_f", c1_))
    return x * torch.sigmoid(c4) * y


def mv_dual_att_alignment_forward(sd, x, extra, pred, flow, max_residue_magnitude=10.0, dg=16):
    cv = lambda k, t, p=0: F.conv2d(t, sd[k + ".weight"], sd.get(k + ".bias"), padding=p)  # noqa: E731

    def gate(z):
        return torch.sigmoid(cv("conv_du.2", F.relu(cv("conv_du.0", z.mean((2, 3), keepdim=True)))))

    warped = flow_warp(extra, flow.permute(0, 2, 3, 1))
    k = cv("fusion_out", torch.cat([warped, pred], 1))
    heads = []
    for v in (warped, pred):
        o = cv("project_out", _channel_attention(x, k, v * gate(v), 8, sd["temperature"]))
        heads.append(cv("conv_offset.2", F.leaky_relu(cv("conv_offset.0", o, 1), 0.1), 1))
    third = 9 * dg
    off = sum(max_residue_magnitude * torch.tanh(h[:, :2 * third]) for h in heads)
    off = off + flow.flip(1).repeat(1, third, 1, 1)
    mask = torch.sigmoid(heads[0][:, 2 * third:] + heads[1][:, 2 * third:])
    return dcn_torch(x, off, mask, sd["weight"], sd.get("bias"), 1, 1, 1, 1, dg)


def seeded_state(shapes, seed, scale=0.15):
    """Deterministic weights for a module given {key: shape} (numpy RandomState stream, sorted keys)."""
    rs = np.random.RandomState(seed)
    sd = {}
    for k in sorted(shapes):
        shp = tuple(shapes[k])
        a = rs.standard_normal(shp).astype(np.float32) * scale
        if k.endswith("temperature"):
            a = 1.0 + a
        sd[k] = torch.from_numpy(a)
    return sd


def seeded_inputs_dsta(B, H, W, seed):
    rs = np.random.RandomState(seed)
    return torch.from_numpy(rs.standard_normal((B, 64, H, W)).astype(np.float32))


def seeded_inputs_mvalign(B, H, W, seed):
    rs = np.random.RandomState(seed)
    t = lambda *s: torch.from_numpy(rs.standard_normal(s).astype(np.float32))  # noqa: E731
    x, extra, pred = t(B, 64, H, W), t(B, 64, H, W), t(B, 64, H, W)
    flow = torch.from_numpy((rs.uniform(-3, 3, (B, 2, H, W))).astype(np.float32))
    return x, extra, pred, flow
