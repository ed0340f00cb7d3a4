"""GPU parity of the CVSR_V7 HIP forward (through the C-ABI) against
  (a) golden outputs of the REAL reference class (tests/golden/cvsr_v7_*.npz, oracle/gen_fixtures.py:run_case_v7), and
  (b) the CPU oracle (oracle/cvsr_v7_ref.py) run here on the same seeded inputs, stage by stage.
Tolerance: 1e-3 max-abs in fp32 (BASELINE.json north_star)."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-3
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "cvsr_v7_*.npz")))


def _model(wseed):
    from arch.SIDECVSR_our import CVSR_V7
    from oracle.cvsr_v7_ref import make_state_dict_v7
    sd = make_state_dict_v7(wseed)
    m = CVSR_V7()
    m.load_state_dict(sd, strict=True)
    return m.cuda().eval(), sd


def _nchw(t):
    return t.permute(0, 3, 1, 2).cpu()


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "fp16x2"])
@pytest.mark.parametrize("path", GOLD, ids=lambda p: os.path.basename(p)[8:-4])
def test_v7_hip_forward_matches_reference_golden(path, precision):
    from oracle.cvsr_v7_ref import cvsr_v7_forward, make_inputs_v7
    g = np.load(path)
    B, H, W = int(g["B"]), int(g["H"]), int(g["W"])
    model, sd = _model(int(g["wseed"]))
    inp = make_inputs_v7(B, H, W, int(g["iseed"]), str(g["layout"]))
    pre = torch.from_numpy(g["pre_L1_fea"]) if int(g["cached"]) else None
    dev = {k: v.cuda() for k, v in inp.items() if k != "gumbel_u"}
    noise = [u.cuda() for u in inp["gumbel_u"]]
    model.debug_taps = {}
    model.precision = precision
    with torch.no_grad():
        out, L1 = model(dev["x"], dev["mvs0"], dev["mvs1"], dev["pms"], dev["rms"], dev["ufs"],
                        None if pre is None else pre.cuda(), gumbel_uniform=noise)
    torch.cuda.synchronize()
    taps = {}
    with torch.no_grad():
        cvsr_v7_forward(sd, inp["x"], inp["mvs0"], inp["mvs1"], inp["pms"], inp["rms"], inp["ufs"], pre, inp["gumbel_u"], taps)
    report, bad = [], []
    for k in ("fused_L3", "fused_L2", "fused_L1", "trunk_L1"):
        e = (_nchw(model.debug_taps[k]) - taps[k]).abs().max().item()
        report.append(f"{k}={e:.2e}")
        if e > TOL * max(1.0, taps[k].abs().max().item()):
            bad.append(k)
    err_out = (out.cpu() - torch.from_numpy(g["out"])).abs().max().item()
    err_l1 = (L1.cpu() - torch.from_numpy(g["L1_fea"])).abs().max().item()
    report += [f"L1_fea={err_l1:.2e}", f"out_vs_golden={err_out:.2e}"]
    print(precision, " ".join(report))
    assert not bad and err_out <= TOL and err_l1 <= TOL, " ".join(report) + f" bad={bad}"
    assert out.shape == (B, 1, 4 * H, 4 * W) and L1.shape == (B * 7, 64, H, W)


def test_v7_surface_and_errors():
    from arch.SIDECVSR_our import CVSR_V7
    from oracle.cvsr_v7_ref import state_dict_spec_v7
    m = CVSR_V7()
    sd = m.state_dict()
    spec = {k: tuple(s) for k, s, *_ in state_dict_spec_v7()}
    assert {k: tuple(v.shape) for k, v in sd.items()} == spec
    with pytest.raises(NotImplementedError):
        m(torch.zeros(1, 7, 1, 8, 8), None, None, None, None, None)
    m = m.cuda().eval()
    z = lambda *s: torch.zeros(*s, device="cuda")  # noqa: E731
    with torch.no_grad(), pytest.raises(ValueError):
        m(z(1, 7, 1, 10, 8), z(1, 7, 2, 10, 8), z(1, 7, 2, 10, 8), z(1, 7, 1, 10, 8), z(1, 1, 7, 10, 8), z(1, 1, 7, 10, 8))
    with pytest.raises(NotImplementedError):     # grad-enabled call: no autograd fallback
        m(z(1, 7, 1, 8, 8), z(1, 7, 2, 8, 8), z(1, 7, 2, 8, 8), z(1, 7, 1, 8, 8), z(1, 1, 7, 8, 8), z(1, 1, 7, 8, 8))


def test_v7_ops_against_torch():
    """The pixel-local V7 operators against their torch definitions (arch.py:1883-1885, 2719-2730, 2836-2845, 4296-4303)."""
    import torch.nn.functional as F
    from cdfo_amd import kernels as K
    g = torch.Generator(device="cuda").manual_seed(11)
    B, H, W = 2, 12, 20
    x = torch.randn(B, H, W, 64, device="cuda", generator=g)
    xn = x.permute(0, 3, 1, 2)
    pooled = torch.cat([xn.max(1, keepdim=True)[0], xn.mean(1, keepdim=True)], 1)
    assert (K.chan_pool(x).permute(0, 3, 1, 2) - pooled).abs().max().item() < 1e-6
    for ks in (7, 3):
        w = torch.randn(1, 2, ks, ks, device="cuda", generator=g) * 0.3
        b = torch.randn(1, device="cuda", generator=g)
        ref = xn * torch.sigmoid(F.conv2d(pooled, w, b, padding=ks // 2))
        assert (K.spatial_gate(x, w, b).permute(0, 3, 1, 2) - ref).abs().max().item() < 1e-5
    xf = torch.randn(B, H, W, 64, device="cuda", generator=g)
    v = torch.rand(B, 64, device="cuda", generator=g) * 3
    u = torch.rand(B, 64, H, W, device="cuda", generator=g).clamp_min_(1e-6)
    w3 = torch.randn(1, 2, 3, 3, device="cuda", generator=g) * 0.3
    b3 = torch.randn(1, device="cuda", generator=g)
    rm = (v.view(B, 64, 1, 1) - (-u.log()).log()).softmax(1)
    ref = xf.permute(0, 3, 1, 2) * (rm + torch.sigmoid(F.conv2d(pooled, w3, b3, padding=1)))
    assert (K.rdab_mix(xf, x, w3, b3, v, u).permute(0, 3, 1, 2) - ref).abs().max().item() < 2e-5
    t = torch.randn(B, 7, 2, 16, 24, device="cuda", generator=g)
    for lv in (1, 2):
        ref = F.interpolate(t[:, 3], scale_factor=0.5 ** lv, mode="bilinear", align_corners=False) / 2 ** lv
        assert (K.shrink_planes(t[:, 3], lv) - ref).abs().max().item() < 1e-6
    a, bb, c = (torch.randn(3, 5, 7, 64, device="cuda", generator=g) for _ in range(3))
    assert (K.lincomb(a, 1.0, bb, 2.0) - (a + 2 * bb)).abs().max().item() < 1e-6
    o = a.clone()
    K.lincomb(o, 1.0, bb, -0.5, c, 3.0, out=o)
    assert (o - (a - 0.5 * bb + 3 * c)).abs().max().item() < 1e-5
