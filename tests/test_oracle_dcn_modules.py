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
