"""CPU: the oracle restatements of DSTA / MVDualAttAlignment against golden vectors from the REAL reference classes
(run by oracle/gen_fixtures.py with only their deformable-conv call replaced by the C oracle)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle.dcn_modules_ref import (dsta_forward, mv_dual_att_alignment_forward, seeded_inputs_dsta,
                                    seeded_inputs_mvalign, seeded_state)

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "dsta_*.npz")) +
              glob.glob(os.path.join(os.path.dirname(__file__), "golden", "mvalign_*.npz")))


def load_case(path):
    g = np.load(path)
    shapes = {k: eval(s) for k, s in zip(g["keys"], g["shapes"])}
    sd = seeded_state(shapes, int(g["wseed"]))
    B, H, W = int(g["B"]), int(g["H"]), int(g["W"])
    if str(g["kind"]) == "dsta":
        inputs = (seeded_inputs_dsta(B, H, W, int(g["iseed"])),)
    else:
        inputs = seeded_inputs_mvalign(B, H, W, int(g["iseed"]))
    return str(g["kind"]), sd, inputs, torch.from_numpy(g["out"])


@pytest.mark.parametrize("path", GOLD, ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_modules_match_reference_golden(path):
    kind, sd, inputs, gold = load_case(path)
    with torch.no_grad():
        out = dsta_forward(sd, *inputs) if kind == "dsta" else mv_dual_att_alignment_forward(sd, *inputs)
    err = (out - gold).abs().max().item()
    assert err <= 2e-5, err


def test_oracle_dcn_node_is_differentiable_and_consistent_with_finite_differences():
    """The gradient oracle of the consumer modules: the C oracle's DCN backward as a torch autograd node (float64).  A directional
    derivative of a scalar loss through DSTA's restatement matches a central finite difference.  The direction moves only the
    parameters at and behind the deformable convolution (its weight / bias, conv4, conv_f, conv_du): the operator's gradients w.r.t.
    offsets and input follow the REFERENCE's backward (deform_conv_cuda_kernel.cu:498-567, 634-766), whose validity rules at the image
    border are not the exact derivative of its forward -- a finite difference through the offset-predicting convolutions differs from it
    by ~1e-3 -- and that convention is pinned elsewhere (tests/test_dcn_oracle.py)."""
    path = [p for p in GOLD if "dsta" in os.path.basename(p)][0]
    _, sd, inputs, gold = load_case(path)
    sd64 = {k: v.double().clone().requires_grad_(True) for k, v in sd.items()}
    x = inputs[0].double()
    cot = torch.from_numpy(np.random.RandomState(1).standard_normal(tuple(gold.shape)))
    loss = lambda s: (dsta_forward(s, x) * cot).sum()  # noqa: E731
    loss(sd64).backward()
    rs = np.random.RandomState(2)
    behind = ("dcn.", "conv4.", "conv_f.", "conv_du.")
    direction = {k: torch.from_numpy(rs.standard_normal(tuple(v.shape))) * float(k.startswith(behind)) for k, v in sd.items()}
    analytic = sum((sd64[k].grad * direction[k]).sum().item() for k in sd)
    eps = 1e-6
    with torch.no_grad():
        lp = loss({k: sd[k].double() + eps * direction[k] for k in sd}).item()
        lm = loss({k: sd[k].double() - eps * direction[k] for k in sd}).item()
    numeric = (lp - lm) / (2 * eps)
    assert abs(analytic - numeric) <= 1e-5 * max(1.0, abs(numeric)), (analytic, numeric)
