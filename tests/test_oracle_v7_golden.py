"""The CPU oracle of CVSR_V7 (oracle/cvsr_v7_ref.py) against golden vectors produced by the REAL reference class
(oracle/gen_fixtures.py:run_case_v7, run in the build container).  This is what pins that oracle."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle.cvsr_v7_ref import N_DRAWS, cvsr_v7_forward, make_inputs_v7, make_state_dict_v7, state_dict_spec_v7


def _cases(golden_dir=os.path.join(os.path.dirname(__file__), "golden")):
    return sorted(glob.glob(os.path.join(golden_dir, "cvsr_v7_*.npz")))


def test_state_dict_has_247_entries_and_param_count():
    spec = state_dict_spec_v7()
    assert len(spec) == 247 and len({k for k, *_ in spec}) == 247
    assert sum(int(np.prod(s)) for _, s, *_ in spec) == 7_491_627      # CVSR_V7().state_dict() of the reference


@pytest.mark.parametrize("path", _cases(), ids=lambda p: os.path.basename(p)[8:-4])
def test_v7_oracle_matches_reference_golden(path):
    g = np.load(path)
    B, H, W = int(g["B"]), int(g["H"]), int(g["W"])
    sd = make_state_dict_v7(int(g["wseed"]))
    inp = make_inputs_v7(B, H, W, int(g["iseed"]), str(g["layout"]))
    assert len(inp["gumbel_u"]) == N_DRAWS
    pre = torch.from_numpy(g["pre_L1_fea"]) if int(g["cached"]) else None
    taps = {}
    with torch.no_grad():
        out, L1 = cvsr_v7_forward(sd, inp["x"], inp["mvs0"], inp["mvs1"], inp["pms"], inp["rms"], inp["ufs"], pre,
                                  inp["gumbel_u"], taps)
    assert out.shape == (B, 1, 4 * H, 4 * W) and L1.shape == (B * 7, 64, H, W)
    assert (L1 - torch.from_numpy(g["L1_fea"])).abs().max().item() <= 1e-4
    for lv in (1, 2, 3):     # hooks on tsa_fusion fire coarse to fine: call 0 = level 3
        ref = torch.from_numpy(g[f"tap_tsa_fusion_{3 - lv}_sample"])
        pre_act = taps[f"fused_L{lv}"]
        got = torch.where(pre_act >= 0, pre_act, pre_act / 0.1).flatten()[::61]       # the hook sees the conv output
        assert (got - ref).abs().max().item() <= 2e-4, lv
    assert (taps["trunk_L1"] - torch.from_numpy(g["trunk_L1"])).abs().max().item() <= 2e-4
    err = (out - torch.from_numpy(g["out"])).abs().max().item()
    assert err <= 5e-5, f"out max-abs {err}"


def test_v7_oracle_reads_both_motion_fields():
    sd = make_state_dict_v7(3)
    inp = make_inputs_v7(1, 8, 8, 7)
    run = lambda m0, m1: cvsr_v7_forward(sd, inp["x"], m0, m1, inp["pms"], inp["rms"], inp["ufs"], None, inp["gumbel_u"])[0]  # noqa: E731
    with torch.no_grad():
        a = run(inp["mvs0"], inp["mvs1"])
        assert not torch.equal(a, run(inp["mvs0"] + 0.7, inp["mvs1"]))
        assert not torch.equal(a, run(inp["mvs0"], inp["mvs1"] + 0.7))
