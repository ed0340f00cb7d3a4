"""Pins the oracle as a GRADIENT oracle: torch autograd through the CPU restatement (oracle/cvsr_v8_ref.py) against the
gradients of the REAL reference's training step (tests/golden/grad_cvsr_v8_*.npz, produced by oracle/gen_fixtures.py:
model.train(); sr, _ = model(...); CharbonnierLoss(sr, hr).backward() -- train_LD_37.py:376-381, opt/loss.py:20-31)."""
import glob
import os

import numpy as np
import pytest
import torch

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "grad_cvsr_v8_*.npz")))


def oracle_grads(g, dtype=torch.float32):
    from oracle.cvsr_v8_ref import cvsr_v8_forward, make_inputs, make_state_dict
    B, H, W = int(g["B"]), int(g["H"]), int(g["W"])
    sd = {k: v.clone().to(dtype).requires_grad_(True) for k, v in make_state_dict(int(g["wseed"])).items()}
    inp = {k: ([u.to(dtype) for u in v] if k == "gumbel_u" else v.to(dtype)) for k, v in make_inputs(B, H, W, int(g["iseed"]), "b1n").items()}
    hr = torch.from_numpy(np.random.RandomState(int(g["hr_seed"])).uniform(0, 1, (B, 1, 4 * H, 4 * W)).astype(np.float32)).to(dtype)
    out, _ = cvsr_v8_forward(sd, inp["x"], None, inp["mvs1"], inp["pms"], inp["rms"], inp["ufs"], None, inp["gumbel_u"])
    d = out - hr
    loss = torch.sum(torch.sqrt(d * d + 1e-4))
    loss.backward()
    return out.detach(), loss.item(), {k: v.grad for k, v in sd.items()}


def golden_errors(g, grads):
    """Per parameter the reference gave a gradient: (max |dg| / max |g|, key) over the strided sample (and the full tensor
    where stored).  Parameters the reference left without gradient (fusion_in) must have none / zeros here too."""
    none = set(g["none"].tolist())
    stride = int(g["stride"])
    rows = []
    for key in [k[2:] for k in g.files if k.startswith("m:")]:
        want_s = g["s:" + key]
        scale = max(float(np.abs(want_s).max()), float(g["m:" + key][4]), -float(g["m:" + key][3]))
        got = grads.get(key)
        if got is None:
            assert scale == 0.0, f"{key}: no gradient produced, reference has max |g| = {scale}"
            continue
        got = got.detach().float().cpu().numpy()
        err = float(np.abs(got.reshape(-1)[::stride] - want_s).max())
        if "f:" + key in g.files:
            err = max(err, float(np.abs(got - g["f:" + key]).max()))
        if scale == 0.0:
            assert err == 0.0, f"{key}: reference gradient is zero, got max {err}"
            continue
        rows.append((err / scale, key))
    for key in none:
        assert grads.get(key) is None or float(grads[key].abs().max()) == 0.0, key
    return sorted(rows, reverse=True)


def compare_with_golden(g, grads, rel):
    """Every tensor within rel * max|g|; returns the worst (relative error, key)."""
    rows = golden_errors(g, grads)
    assert rows[0][0] <= rel, f"{rows[0][1]}: {rows[0][0]:.3e} of max |g|"
    return rows[0]


@pytest.mark.parametrize("path", GOLD, ids=lambda p: os.path.basename(p)[13:-4])
def test_oracle_autograd_matches_reference_gradients(path):
    g = np.load(path)
    out, loss, grads = oracle_grads(g)
    assert np.abs(out.numpy() - g["out"]).max() == 0.0
    assert abs(loss - float(g["loss"])) <= 1e-6 * abs(float(g["loss"]))
    worst = compare_with_golden(g, grads, 2e-5)
    print("oracle vs reference gradients, worst relative error:", worst)
