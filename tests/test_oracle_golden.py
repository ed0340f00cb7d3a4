"""The CPU oracle (oracle/cvsr_v8_ref.py) against golden vectors produced by the REAL reference
(oracle/gen_fixtures.py, run in the build container).  This is what pins the oracle."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle.cvsr_v8_ref import cvsr_v8_forward, make_inputs, make_state_dict, state_dict_spec

TOL = 2e-5  # fp32 re-association only; same ATen kernels on both sides


def _cases(golden_dir=os.path.join(os.path.dirname(__file__), "golden")):
    return sorted(glob.glob(os.path.join(golden_dir, "cvsr_v8_*.npz")))


def test_state_dict_has_261_entries_and_param_count():
    spec = state_dict_spec()
    assert len(spec) == 261
    assert len({k for k, *_ in spec}) == 261
    n = sum(int(np.prod(s)) for _, s, *_ in spec)
    assert n == 7_098_392


@pytest.mark.parametrize("path", _cases(), ids=lambda p: os.path.basename(p)[8:-4])
def test_oracle_matches_reference_golden(path):
    g = np.load(path)
    B, H, W = int(g["B"]), int(g["H"]), int(g["W"])
    sd = make_state_dict(int(g["wseed"]))
    inp = make_inputs(B, H, W, int(g["iseed"]), str(g["layout"]))
    pre = torch.from_numpy(g["pre_L1_fea"]) if int(g["cached"]) else None
    taps = {}
    with torch.no_grad():
        out, L1 = cvsr_v8_forward(sd, inp["x"], inp["mvs0"], inp["mvs1"], inp["pms"], inp["rms"], inp["ufs"],
                                  pre, inp["gumbel_u"], taps)
    assert out.shape == (B, 1, 4 * H, 4 * W) and L1.shape == (B * 7, 64, H, W)
    err = (out - torch.from_numpy(g["out"])).abs().max().item()
    assert err <= TOL, f"out max-abs {err}"
    if "L1_fea" in g:
        e = (L1 - torch.from_numpy(g["L1_fea"])).abs().max().item()
    else:
        e = (L1.flatten()[::97] - torch.from_numpy(g["L1_fea_sample"])).abs().max().item()
    assert e <= 1e-4, f"L1_fea max-abs {e}"
    # stage taps: strided samples of the hooks on RDAB / MV_deform_align / tsa_fusion / recon_trunk
    nbrs = [i for i in range(7) if i != 3]
    for j, i in enumerate(nbrs):
        for nm, key in (("RDAB", f"rdab_{i}"), ("MV_deform_align", f"align_{i}")):
            ref = torch.from_numpy(g[f"tap_{nm}_{j}_sample"])
            got = taps[key].flatten()[::61]
            assert (got - ref).abs().max().item() <= 1e-4, (nm, i)
    ref = torch.from_numpy(g["tap_recon_trunk_0_sample"])
    assert (taps["trunk"].flatten()[::61] - ref).abs().max().item() <= 1e-4


def test_oracle_ignores_mvs0_like_the_reference():
    sd = make_state_dict(3)
    inp = make_inputs(1, 8, 8, 7)
    with torch.no_grad():
        a, _ = cvsr_v8_forward(sd, inp["x"], inp["mvs0"], inp["mvs1"], inp["pms"], inp["rms"], inp["ufs"], None,
                               inp["gumbel_u"])
        b, _ = cvsr_v8_forward(sd, inp["x"], inp["mvs0"] + 5.0, inp["mvs1"], inp["pms"], inp["rms"], inp["ufs"],
                               None, inp["gumbel_u"])
    assert torch.equal(a, b)
