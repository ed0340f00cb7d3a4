"""ORACLE tooling (test infrastructure): generate golden vectors FROM THE REAL REFERENCE.

Runs only in the build container, where ``/root/reference`` is mounted read-only.  It loads the
reference's ``arch/SIDECVSR_our.py`` *as is* (no source is copied), after registering stand-in modules for
the third-party imports that file makes but the V8 path never calls (torchvision, timm, cv2, and the
non-existent ``arch.ops.dcn`` of arch.py:9 -- SURVEY section 0/F3, section 8c), neutralising the debug
side effects (``featuremap_visual`` -> no-op, ``nn.Module.cuda`` -> identity for arch.py:2161-2162) and
wrapping ``torch.rand_like`` so the six Gumbel draws are the seeded tensors of
``oracle.cvsr_v8_ref.make_inputs``.

Weights and inputs are regenerated from seeds by ``oracle.cvsr_v8_ref`` (numpy ``RandomState`` streams),
so the committed fixtures hold only the reference's OUTPUTS (data, not code):

    tests/golden/cvsr_v8_<case>.npz : out, L1_fea (small cases) or their strided samples + moments,
                                      per-stage taps' moments, the case parameters.

Usage:  python oracle/gen_fixtures.py            (writes tests/golden/*.npz)
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)
from oracle.cvsr_v8_ref import make_inputs, make_state_dict  # noqa: E402

CASES = {
    # name: (B, H, W, weight_seed, input_seed, layout, cached)
    "b1_8x8": (1, 8, 8, 11, 101, "b1n", False),
    "b1_16x16": (1, 16, 16, 12, 102, "b1n", False),
    "b2_16x24": (2, 16, 24, 13, 103, "b1n", False),
    "b2_16x24_bn1": (2, 16, 24, 13, 103, "bn1", False),
    "b1_16x16_cached": (1, 16, 16, 12, 104, "b1n", True),
    "b1_24x40": (1, 24, 40, 14, 105, "b1n", False),
}


def load_reference():
    sys.dont_write_bytecode = True
    import matplotlib
    matplotlib.use("Agg")

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__path__ = []  # behave like a package
        sys.modules[name] = m
        return m

    class _PackStub(nn.Module):
        """Parameter/attribute container standing in for the non-existent ``arch.ops.dcn.ModulatedDeformConvPack``
        (arch.py:9): lets the subclasses at arch.py:3103-3352,3653 be defined and gives ``MVDualAttAlignment`` the
        attributes its forward reads (ops/dcn/deform_conv.py:264-337 layout).  No forward: never called."""
        def __init__(self, in_channels=64, out_channels=64, kernel_size=3, stride=1, padding=0, dilation=1, groups=1,
                     deformable_groups=1, bias=True):
            super().__init__()
            self.in_channels, self.out_channels = in_channels, out_channels
            self.kernel_size = (kernel_size, kernel_size)
            self.stride, self.padding, self.dilation = stride, padding, dilation
            self.groups, self.deformable_groups, self.with_bias = groups, deformable_groups, bias
            self.weight = nn.Parameter(torch.zeros(out_channels, in_channels // groups, kernel_size, kernel_size))
            self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
            self.conv_offset_mask = nn.Conv2d(in_channels, deformable_groups * 3 * kernel_size * kernel_size,
                                              kernel_size, stride, padding)

    saved = {k: sys.modules.get(k) for k in ("arch", "arch.ops", "arch.ops.dcn")}
    from oracle.dcn_modules_ref import dcn_torch

    def _deform_conv2d(input, offset, weight, bias=None, stride=1, padding=0, dilation=1, mask=None):
        # stand-in for the un-vendored torchvision.ops.deform_conv2d (arch.py:3352): the C oracle of the DCN forward
        dg = offset.shape[1] // (2 * weight.shape[2] * weight.shape[3])
        return dcn_torch(input, offset, mask, weight, bias, stride, padding, dilation, 1, dg)

    stub("torchvision"); stub("torchvision.ops", deform_conv2d=_deform_conv2d); stub("torchvision.datasets")
    stub("torchvision.transforms"); stub("torchvision.utils", save_image=lambda *a, **k: None)
    sys.modules["torchvision"].datasets = sys.modules["torchvision.datasets"]
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision"].ops = sys.modules["torchvision.ops"]
    stub("timm"); stub("timm.models")
    stub("timm.models.layers", DropPath=nn.Identity, to_2tuple=lambda v: (v, v),
         trunc_normal_=lambda t, *a, **k: t)
    stub("cv2")
    stub("arch"); stub("arch.ops"); stub("arch.ops.dcn", ModulatedDeformConvPack=_PackStub)
    nn.Module.cuda = lambda self, *a, **k: self

    spec = importlib.util.spec_from_file_location("_cdfo_reference_arch", os.path.join(REF, "arch", "SIDECVSR_our.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    ref.featuremap_visual = lambda *a, **k: None
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v
    return ref


def moments(t: torch.Tensor):
    t = t.double()
    return np.array([t.mean().item(), t.abs().mean().item(), t.std().item(), t.min().item(), t.max().item()])


def run_case(ref, name, B, H, W, wseed, iseed, layout, cached):
    sd = make_state_dict(wseed)
    model = ref.CVSR_V8()
    missing = model.load_state_dict(sd, strict=True)
    model.eval()
    inp = make_inputs(B, H, W, iseed, layout)
    queue = list(inp["gumbel_u"])
    real_rand_like = torch.rand_like

    def fake_rand_like(t, *a, **k):
        u = queue.pop(0)
        assert u.shape == t.shape, (u.shape, t.shape)
        return u

    taps = {}
    hooks = []
    for nm in ("RDAB", "MV_deform_align", "tsa_fusion", "recon_trunk", "conv_last"):
        lst = taps.setdefault(nm, [])
        hooks.append(getattr(model, nm).register_forward_hook(lambda m, i, o, lst=lst: lst.append(o.detach().clone())))

    pre = None
    if cached:
        # previous-call feature cache: run the reference once on a shifted clip to obtain a real L1_fea
        prev = make_inputs(B, H, W, iseed + 1000, layout)
        with torch.no_grad():
            torch.rand_like = lambda t, *a, **k: real_rand_like(t)
            try:
                _, pre = model(prev["x"], prev["mvs0"], prev["mvs1"], prev["pms"], prev["rms"], prev["ufs"])
            finally:
                torch.rand_like = real_rand_like
        for lst in taps.values():
            lst.clear()
    torch.rand_like = fake_rand_like
    try:
        with torch.no_grad():
            out, L1 = model(inp["x"], inp["mvs0"], inp["mvs1"], inp["pms"], inp["rms"], inp["ufs"], pre)
    finally:
        torch.rand_like = real_rand_like
    for h in hooks:
        h.remove()
    assert not queue, "reference drew fewer noise tensors than expected"

    rec = dict(B=B, H=H, W=W, wseed=wseed, iseed=iseed, layout=layout, cached=int(cached),
               out=out.numpy(), L1_moments=moments(L1))
    if L1.numel() <= 130_000:
        rec["L1_fea"] = L1.numpy()
    else:
        rec["L1_fea_sample"] = L1.flatten()[::97].numpy().copy()
    if cached:
        rec["pre_L1_fea"] = pre.numpy()
    for nm, lst in taps.items():
        for j, t in enumerate(lst):
            rec[f"tap_{nm}_{j}_moments"] = moments(t)
            rec[f"tap_{nm}_{j}_sample"] = t.flatten()[::61].numpy().copy()
    path = os.path.join(REPO, "tests", "golden", f"cvsr_v8_{name}.npz")
    np.savez_compressed(path, **rec)
    print(f"{name}: out {tuple(out.shape)} mean {out.mean():.6f} -> {path} ({os.path.getsize(path)/1024:.0f} KiB)")


def load_reference_dsta():
    """ops/attentionlayer.py of the reference, with ``ops.dcn.deform_conv.ModulatedDeformConv`` (needs the stripped
    CUDA extension) replaced by a parameter-compatible module whose forward is the C oracle."""
    from oracle.dcn_modules_ref import dcn_torch

    class _MDC(nn.Module):
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                     deformable_groups=1, bias=True):
            super().__init__()
            self.stride, self.padding, self.dilation, self.groups, self.dg = stride, padding, dilation, groups, deformable_groups
            self.weight = nn.Parameter(torch.zeros(out_channels, in_channels // groups, kernel_size, kernel_size))
            self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None

        def forward(self, x, offset, mask):
            return dcn_torch(x, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation,
                             self.groups, self.dg)

    saved = {k: sys.modules.get(k) for k in ("ops", "ops.dcn", "ops.dcn.deform_conv")}
    for name in ("ops", "ops.dcn"):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    m = types.ModuleType("ops.dcn.deform_conv")
    m.ModulatedDeformConv = _MDC
    sys.modules["ops.dcn.deform_conv"] = m
    spec = importlib.util.spec_from_file_location("_cdfo_reference_attentionlayer", os.path.join(REF, "ops", "attentionlayer.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v
    return mod


MODULE_CASES = {   # name: (kind, B, H, W, weight seed, input seed)
    "dsta_b2_48x64": ("dsta", 2, 48, 64, 41, 141),
    "dsta_b1_40x56": ("dsta", 1, 40, 56, 42, 142),
    "mvalign_b2_16x24": ("mvalign", 2, 16, 24, 43, 143),
    "mvalign_b1_24x40": ("mvalign", 1, 24, 40, 44, 144),
}


def run_module_case(ref, ref_att, name, kind, B, H, W, wseed, iseed):
    from oracle.dcn_modules_ref import seeded_inputs_dsta, seeded_inputs_mvalign, seeded_state
    if kind == "dsta":
        model = ref_att.DSTA(64)
        inputs = (seeded_inputs_dsta(B, H, W, iseed),)
    else:
        model = ref.MVDualAttAlignment(64, 64, 3, padding=1, deformable_groups=16, max_residue_magnitude=10)
        inputs = seeded_inputs_mvalign(B, H, W, iseed)
    sd = seeded_state({k: v.shape for k, v in model.state_dict().items()}, wseed)
    model.load_state_dict(sd, strict=True)
    model.eval()
    with torch.no_grad():
        out = model(*inputs)
    path = os.path.join(REPO, "tests", "golden", f"{name}.npz")
    np.savez_compressed(path, kind=kind, B=B, H=H, W=W, wseed=wseed, iseed=iseed, out=out.numpy(),
                        keys=np.array(sorted(sd)), shapes=np.array([str(tuple(sd[k].shape)) for k in sorted(sd)]))
    print(f"{name}: out {tuple(out.shape)} absmean {out.abs().mean():.5f} -> {path} ({os.path.getsize(path)/1024:.0f} KiB)")


def gen_streaming_helpers():
    """Golden vectors of the reference's streaming-loop helpers (test_LD_22_FPS.py:14-17,100-122,200-225).  The script
    itself cannot be imported (module-level cv2 / dataset paths), so the three FunctionDefs are compiled from its
    source in place; nothing is copied into the repo but their outputs."""
    import ast
    src = open(os.path.join(REF, "test_LD_22_FPS.py")).read()
    wanted = {"generate_input_index", "mv2mvs", "modify_mv_for_end_frames"}
    mod = ast.Module([n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name in wanted], [])
    ns = {"np": np, "torch": torch}
    exec(compile(mod, "test_LD_22_FPS.py", "exec"), ns)
    rs = np.random.RandomState(7)
    rec = {}
    idx_cases = [(c, 7, m) for m in (0, 2, 5, 9) for c in range(m + 1)]
    rec["index_cases"] = np.array(idx_cases)
    rec["index_out"] = np.stack([ns["generate_input_index"](*c) for c in idx_cases])
    mv = rs.randint(-64, 64, size=(3, 6, 10, 3)).astype(np.float32)
    mv[..., 2] = rs.choice([-2.0, -1.0, 0.0, 1.0, 4.0], size=mv.shape[:-1])      # 0 distance: 0/0 -> NaN -> 0, x/0 -> inf
    mv[0, 0, 0] = (0.0, 0.0, 0.0)
    rec["mv_in"] = mv
    with np.errstate(divide="ignore", invalid="ignore"):
        rec["mv_out"] = np.stack([ns["mv2mvs"](m.copy()).numpy() for m in mv])
    mod_cases, mod_out = [], []
    base = rs.randn(1, 7, 2, 3, 4).astype(np.float32)
    for T in (1, 2, 3, 4, 5, 8):
        for i in range(T):
            t = torch.from_numpy(base.copy())
            ns["modify_mv_for_end_frames"](i, t, T)
            mod_cases.append((i, T))
            mod_out.append(t.numpy())
    rec["mod_base"], rec["mod_cases"], rec["mod_out"] = base, np.array(mod_cases), np.stack(mod_out)
    path = os.path.join(REPO, "tests", "golden", "streaming_helpers.npz")
    np.savez_compressed(path, **rec)
    print(f"streaming helpers -> {path} ({os.path.getsize(path)/1024:.0f} KiB)")


def gen_metrics_psnr():
    """Golden vectors of the reference's calculate_psnr (metric/psnr_ssim.py:278-317; numpy only, compiled in place
    from its source -- the module itself imports cv2)."""
    import ast
    src = open(os.path.join(REF, "metric", "psnr_ssim.py")).read()
    mod = ast.Module([n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "calculate_psnr"], [])
    ns = {"np": np}
    exec(compile(mod, "psnr_ssim.py", "exec"), ns)
    rs = np.random.RandomState(3)
    a = rs.randint(0, 256, size=(4, 40, 56, 1)).astype(np.float64)
    b = np.clip(a + np.round(rs.randn(*a.shape) * np.array([0.0, 1.0, 4.0, 20.0])[:, None, None, None]), 0, 255)
    out = np.array([[ns["calculate_psnr"](x, y, c) for c in (0, 4)] for x, y in zip(a, b)])
    path = os.path.join(REPO, "tests", "golden", "metrics_psnr.npz")
    np.savez_compressed(path, a=a.astype(np.uint8), b=b.astype(np.uint8), crops=np.array([0, 4]), psnr=out)
    print(f"metrics psnr -> {path}: {out.tolist()}")


V7_CASES = {   # name: (B, H, W, weight seed, input seed, layout, cached)
    "b1_16x16": (1, 16, 16, 21, 201, "b1n", False),
    "b2_16x24_bn1": (2, 16, 24, 22, 202, "bn1", False),
    "b1_24x32_cached": (1, 24, 32, 23, 203, "b1n", True),
}


def run_case_v7(ref, name, B, H, W, wseed, iseed, layout, cached):
    """The REAL ``CVSR_V7`` (arch.py:4215-4367) on seeded weights / inputs; its RDAB noise is fed from the seeded list
    by intercepting ``torch.rand_like`` and its ``torchvision.ops.deform_conv2d`` is the C oracle (load_reference)."""
    from oracle.cvsr_v7_ref import make_inputs_v7, make_state_dict_v7
    sd = make_state_dict_v7(wseed)
    model = ref.CVSR_V7()
    model.load_state_dict(sd, strict=True)
    model.eval()
    inp = make_inputs_v7(B, H, W, iseed, layout)
    queue = list(inp["gumbel_u"])
    real_rand_like = torch.rand_like

    def fake_rand_like(t, *a, **k):
        u = queue.pop(0)
        assert u.shape == t.shape, (u.shape, t.shape)
        return u

    taps = {}
    hooks = []
    for nm in ("transformer_feature_extraction", "RDAB", "MV_deform_align", "fb_fusion", "tsa_fusion"):
        lst = taps.setdefault(nm, [])
        hooks.append(getattr(model, nm).register_forward_hook(lambda m, i, o, lst=lst: lst.append(o.detach().clone())))
    trunk = []
    hooks.append(model.recon_trunk.register_forward_hook(lambda m, i, o: trunk.extend(t.detach().clone() for t in o)))
    pre = None
    if cached:
        prev = make_inputs_v7(B, H, W, iseed + 1000, layout)
        with torch.no_grad():
            torch.rand_like = lambda t, *a, **k: real_rand_like(t)
            try:
                _, pre = model(prev["x"], prev["mvs0"], prev["mvs1"], prev["pms"], prev["rms"], prev["ufs"])
            finally:
                torch.rand_like = real_rand_like
        for lst in taps.values():
            lst.clear()
        trunk.clear()
    torch.rand_like = fake_rand_like
    try:
        with torch.no_grad():
            out, L1 = model(inp["x"], inp["mvs0"], inp["mvs1"], inp["pms"], inp["rms"], inp["ufs"], pre)
    finally:
        torch.rand_like = real_rand_like
    for h in hooks:
        h.remove()
    assert not queue, "reference drew fewer noise tensors than expected"
    rec = dict(B=B, H=H, W=W, wseed=wseed, iseed=iseed, layout=layout, cached=int(cached), out=out.numpy(),
               L1_moments=moments(L1), L1_fea=L1.numpy())
    if cached:
        rec["pre_L1_fea"] = pre.numpy()
    for j, t in enumerate(trunk):
        rec[f"trunk_L{j + 1}"] = t.numpy()
    for nm, lst in taps.items():
        for j, t in enumerate(lst):
            rec[f"tap_{nm}_{j}_moments"] = moments(t)
            rec[f"tap_{nm}_{j}_sample"] = t.flatten()[::61].numpy().copy()
    path = os.path.join(REPO, "tests", "golden", f"cvsr_v7_{name}.npz")
    np.savez_compressed(path, **rec)
    print(f"v7 {name}: out {tuple(out.shape)} mean {out.mean():.6f} -> {path} ({os.path.getsize(path)/1024:.0f} KiB)")


GRAD_CASES = {
    # name: (B, H, W, weight_seed, input_seed)   -- training-mode gradients of the REAL reference (train_LD_37.py:376-381)
    "grad_b1_8x8": (1, 8, 8, 21, 201),
    "grad_b2_16x16": (2, 16, 16, 22, 202),
}
GRAD_STRIDE = 53          # every parameter's gradient is stored as its moments + every 53rd element; small ones in full


def charbonnier(x, y):
    """opt/loss.py:20-31 as train_LD_37.py:377 calls it: sum(sqrt(diff^2 + 1e-4))."""
    d = x - y
    return torch.sum(torch.sqrt(d * d + 1e-4))


def run_grad_case(ref, name, B, H, W, wseed, iseed):
    """One training step's forward + backward of the real reference class: model.train(); sr, _ = model(...);
    loss = CharbonnierLoss(sr, hr); loss.backward()  -- the outputs kept are the loss, `out`, and every parameter's .grad
    (moments + strided sample; in full when it has at most 4096 elements)."""
    sd = make_state_dict(wseed)
    model = ref.CVSR_V8()
    model.load_state_dict(sd, strict=True)
    model.train()
    inp = make_inputs(B, H, W, iseed, "b1n")
    hr = torch.from_numpy(np.random.RandomState(iseed + 7).uniform(0, 1, (B, 1, 4 * H, 4 * W)).astype(np.float32))
    queue = list(inp["gumbel_u"])
    real_rand_like = torch.rand_like

    def fake_rand_like(t, *a, **k):
        u = queue.pop(0)
        assert u.shape == t.shape
        return u

    torch.rand_like = fake_rand_like
    try:
        out, _ = model(inp["x"], inp["mvs0"], inp["mvs1"], inp["pms"], inp["rms"], inp["ufs"])
    finally:
        torch.rand_like = real_rand_like
    loss = charbonnier(out, hr)
    loss.backward()
    rec = dict(B=B, H=H, W=W, wseed=wseed, iseed=iseed, out=out.detach().numpy(), loss=np.float64(loss.item()), hr_seed=iseed + 7,
               stride=GRAD_STRIDE)
    none = []
    for k, prm in model.named_parameters():
        if prm.grad is None:
            none.append(k)
            continue
        g = prm.grad.detach()
        rec["m:" + k] = moments(g)
        rec["s:" + k] = g.flatten()[::GRAD_STRIDE].numpy().copy()
        if g.numel() <= 4096:
            rec["f:" + k] = g.numpy().copy()
    rec["none"] = np.array(none)
    path = os.path.join(REPO, "tests", "golden", f"grad_cvsr_v8_{name[5:]}.npz")
    np.savez_compressed(path, **rec)
    print(f"{name}: loss {loss.item():.6f}, {len(none)} parameters without gradient {none[:4]} -> {path} "
          f"({os.path.getsize(path)/1024:.0f} KiB)")


def main():
    if sys.argv[1:] == ["streaming"]:
        return gen_streaming_helpers()
    if sys.argv[1:] == ["metrics"]:
        return gen_metrics_psnr()
    ref = load_reference()
    only = sys.argv[1:]
    for name, cfg in CASES.items():
        if only and name not in only:
            continue
        run_case(ref, name, *cfg)
    for name, cfg in GRAD_CASES.items():
        if only and name not in only:
            continue
        run_grad_case(ref, name, *cfg)
    for name, cfg in V7_CASES.items():
        if only and "v7_" + name not in only:
            continue
        run_case_v7(ref, name, *cfg)
    ref_att = load_reference_dsta()
    for name, cfg in MODULE_CASES.items():
        if only and name not in only:
            continue
        run_module_case(ref, ref_att, name, *cfg)
    if not only:
        gen_streaming_helpers()
        gen_metrics_psnr()


if __name__ == "__main__":
    main()
