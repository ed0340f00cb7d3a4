"""GPU parity of the gather / resampling operators against their torch definitions, including the motion fields the
reference's own loader can produce (mv2mvs leaves x / 0 = inf, test_LD_22_FPS.py:106-110)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H,W", [(2, 16, 24), (1, 9, 31), (1, 8, 8)])
def test_flow_warp_matches_grid_sample_and_survives_non_finite_flows(B, H, W):
    from cdfo_amd import kernels as K
    from oracle.cvsr_v8_ref import flow_warp as ref_warp
    g = torch.Generator().manual_seed(B * 100 + H)
    x = torch.randn(B, 64, H, W, generator=g)
    flow = torch.randn(B, 2, H, W, generator=g) * 3.0                 # channel 0 = x, 1 = y, pixels; some leave the image
    flow[0, 0, 0, 0], flow[0, 1, 1, 1], flow[0, 0, 2, 2] = 1e9, -1e9, -float(W)
    ref = ref_warp(x, flow.permute(0, 2, 3, 1))
    out = K.flow_warp(K.nchw_to_nhwc(x.cuda()), flow.cuda().contiguous(), 2 * H * W)
    torch.cuda.synchronize()
    assert (out.permute(0, 3, 1, 2).cpu() - ref).abs().max().item() < 1e-5
    bad = flow.clone()
    bad[0, 0, 3, 3], bad[0, 1, 4, 4], bad[0, 0, 5, 5] = float("inf"), float("-inf"), float("nan")
    out2 = K.flow_warp(K.nchw_to_nhwc(x.cuda()), bad.cuda().contiguous(), 2 * H * W).permute(0, 3, 1, 2).cpu()
    torch.cuda.synchronize()
    assert out2[0, :, 3, 3].abs().max().item() == 0.0 and out2[0, :, 4, 4].abs().max().item() == 0.0      # no corner in range
    assert out2[0, :, 5, 5].abs().max().item() == 0.0
    keep = torch.ones(B, 1, H, W, dtype=torch.bool)
    keep[0, 0, 3, 3] = keep[0, 0, 4, 4] = keep[0, 0, 5, 5] = False
    assert ((out2 - ref).abs() * keep).max().item() < 1e-5
