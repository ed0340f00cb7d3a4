"""The CVSR_V8 module under autograd on the GPU (SURVEY section 8f n2 / boundary B1): train_LD_37.py:376-381's call pattern
-- model.train(); optimizer.zero_grad(); sr, _ = model(...); loss = CharbonnierLoss(sr, hr); loss.backward();
optimizer.step() -- with HIP kernels forward and backward, against

  * the gradients of the REAL reference's training step (tests/golden/grad_cvsr_v8_*.npz): every parameter; the forward
    `out` <= 1e-5 (exact-fp32 kernels);
  * the CPU oracle's autograd at another size / batch (pinned to the reference by tests/test_oracle_grad_golden.py), in
    float64 as the truth and in float32 as the yardstick.

Tolerances.  The typical HIP gradient tensor agrees with the reference to fp32 rounding (median 3e-7 of max |g|, the same as the
CPU restatement's own float32-vs-float64 difference).  Individual tensors can be off by ~1e-3: ReLU / LeakyReLU / arg-max
derivatives are discontinuous, and when ONE activation sits within rounding of its kink, two correct fp32 implementations
take different sides (measured: at 16x16, B = 2 the prior U-net's tensors 2-6e-3 with everything else at 1e-6; at 32x32 one
trunk convolution 9e-4; the CPU restatement in float32 shows the same effect against float64 elsewhere: 5.8e-3 on
`conv_expand_rms.weight` at another seed; the prior path's gradients are small differences of large contributions, so the
same noise weighs more there).  Hence per case: median <= 5e-6, 75th percentile <= 1e-3, 90th percentile <= 5e-3, every tensor
<= 2e-2 of its max |g|; at 8x8 (few kinks) every tensor <= 1e-3."""
import glob
import os

import numpy as np
import pytest
import torch

from test_oracle_grad_golden import golden_errors, oracle_grads

pytestmark = pytest.mark.gpu
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "grad_cvsr_v8_*.npz")))


def _charbonnier(x, y):
    d = x - y
    return torch.sum(torch.sqrt(d * d + 1e-4))


def _hip_step(wseed, inp, hr, noise):
    from arch.SIDECVSR_our import CVSR_V8
    from oracle.cvsr_v8_ref import make_state_dict
    m = CVSR_V8()
    m.load_state_dict(make_state_dict(wseed), strict=True)
    m = m.cuda().train()
    d = {k: v.cuda() for k, v in inp.items() if k != "gumbel_u"}
    out, L1 = m(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"], gumbel_uniform=noise)
    loss = _charbonnier(out, hr.cuda())
    loss.backward()
    torch.cuda.synchronize()
    return m, out.detach().cpu(), loss.item(), {k: p.grad for k, p in m.named_parameters()}


@pytest.fixture(params=["exact", "default"])
def conv_mode(request):
    """The training path's two convolution arithmetics (cdfo_amd.autograd.CONV_PREC): exact-fp32 MFMA -- what the tolerances
    below were derived for -- and the round-3 default, split-bf16 three-pass MFMA for the 3x3 convolutions (fp32-grade products;
    the operands keep ~16 bits, so the median gradient error moves from ~3e-7-1e-6 to 1e-5-4e-5 of max |g| -- two orders below what
    bf16 mixed-precision training works with, and below the kink noise that sets the upper percentiles; bound here: 1e-4)."""
    from cdfo_amd import autograd as A
    from cdfo_amd import kernels as K
    old = A.CONV_PREC
    A.CONV_PREC = K.PREC_F32 if request.param == "exact" else K.PREC_BF16X3
    yield request.param
    A.CONV_PREC = old


@pytest.mark.parametrize("path", GOLD, ids=lambda p: os.path.basename(p)[13:-4])
def test_hip_training_step_matches_reference_gradients(path, conv_mode):
    from oracle.cvsr_v8_ref import make_inputs
    exact = conv_mode == "exact"
    g = np.load(path)
    B, H, W = int(g["B"]), int(g["H"]), int(g["W"])
    inp = make_inputs(B, H, W, int(g["iseed"]), "b1n")
    hr = torch.from_numpy(np.random.RandomState(int(g["hr_seed"])).uniform(0, 1, (B, 1, 4 * H, 4 * W)).astype(np.float32))
    m, out, loss, grads = _hip_step(int(g["wseed"]), inp, hr, [u.cuda() for u in inp["gumbel_u"]])
    assert np.abs(out.numpy() - g["out"]).max() <= (1e-5 if exact else 3e-5)
    assert abs(loss - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    rows = golden_errors(g, grads)
    errs = np.array([r[0] for r in rows])
    print(f"HIP training step ({conv_mode} convolutions) vs the reference's gradients ({os.path.basename(path)}): {len(rows)} tensors, "
          f"median {np.median(errs):.2e}, 90th percentile {np.quantile(errs, 0.9):.2e}, worst {rows[0][0]:.2e} ({rows[0][1]})")
    assert np.median(errs) <= (5e-6 if exact else 1e-4) and np.quantile(errs, 0.75) <= 1e-3 and np.quantile(errs, 0.9) <= 5e-3 and errs.max() <= 2e-2
    if H * W <= 64:
        assert errs.max() <= 1e-3


def test_hip_training_step_matches_oracle_autograd_at_another_size(conv_mode):
    """B = 2 clips of 24x16 (non-square, two images per launch: per-image attention weights, batch strides).  Truth = the
    oracle's float64 autograd; the HIP (fp32) gradients must be as close to it as the oracle's own float32 gradients are."""
    from oracle.cvsr_v8_ref import make_inputs
    fake = dict(B=2, H=24, W=16, wseed=31, iseed=301, hr_seed=308, stride=53)
    out_o, loss_o, g64 = oracle_grads(fake, torch.float64)
    _, _, g32 = oracle_grads(fake, torch.float32)
    inp = make_inputs(2, 24, 16, 301, "b1n")
    hr = torch.from_numpy(np.random.RandomState(308).uniform(0, 1, (2, 1, 96, 64)).astype(np.float32))
    m, out, loss, grads = _hip_step(31, inp, hr, [u.cuda() for u in inp["gumbel_u"]])
    exact = conv_mode == "exact"
    assert (out.double() - out_o).abs().max().item() <= (1e-5 if exact else 3e-5)
    rows = []
    for k, go in g64.items():
        if go is None:
            assert grads[k] is None or grads[k].abs().max().item() == 0.0, k
            continue
        scale = go.abs().max().item()
        if scale == 0.0:
            continue
        rows.append(((grads[k].cpu().double() - go).abs().max().item() / scale, (g32[k].double() - go).abs().max().item() / scale, k))
    rows.sort(reverse=True)
    e_hip, e_cpu = np.array([r[0] for r in rows]), np.array([r[1] for r in rows])
    print(f"24x16, B=2 ({conv_mode} convolutions) vs float64 oracle autograd: HIP median {np.median(e_hip):.2e} (float32 CPU oracle {np.median(e_cpu):.2e}), "
          f"90th percentile {np.quantile(e_hip, 0.9):.2e}, worst {rows[0][0]:.2e} ({rows[0][2]})")
    assert np.median(e_hip) <= (max(5e-6, 3 * np.median(e_cpu)) if exact else 1e-4) and np.quantile(e_hip, 0.75) <= 1e-3 and np.quantile(e_hip, 0.9) <= 5e-3 \
        and e_hip.max() <= 2e-2


def test_train_script_call_pattern_runs_and_learns():
    """train_LD_37.py:324-381: Adam, model.train(), the six-argument call, Charbonnier loss, backward, step -- three steps on
    one batch lower the loss; the default noise path (drawn in the kernel) is used, as the script does not pass any."""
    from arch.SIDECVSR_our import CVSR_V8
    from oracle.cvsr_v8_ref import make_inputs, make_state_dict
    torch.manual_seed(0)
    model = CVSR_V8(SCGs=8)
    model.load_state_dict(make_state_dict(41), strict=True)
    model = model.cuda()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=0.0)
    model.train()
    inp = make_inputs(2, 16, 16, 401, "b1n")
    d = {k: v.cuda() for k, v in inp.items() if k != "gumbel_u"}
    hr = torch.rand(2, 1, 64, 64, device="cuda")
    losses = []
    for _ in range(3):
        opt.zero_grad()
        torch.manual_seed(5)                      # same noise draws every step, so that the loss is comparable
        sr, _ = model(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"])
        loss = _charbonnier(sr, hr)
        losses.append(loss.item())
        loss.backward()
        opt.step()
    assert all(np.isfinite(losses)) and losses[2] < losses[0], losses
    model.eval()
    with torch.no_grad():                         # and the trained weights run through the inference schedule
        out, _ = model(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"])
    assert torch.isfinite(out).all()
