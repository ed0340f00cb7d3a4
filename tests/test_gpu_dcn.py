"""GPU parity of the fused HIP DCN forward (through the reference-shaped Python operator surface and the C-ABI)
against the C oracle (oracle/dcn_ref.c) on the same seeded inputs; plus the reference's DCNv1 known-answer test."""
import numpy as np
import pytest
import torch

from dcn_oracle import dcn_forward_ref

pytestmark = pytest.mark.gpu


def test_simple_check_kat_through_the_module_surface():
    from ops.dcn.deform_conv import DeformConv
    m = DeformConv(2, 1, kernel_size=3, padding=1, deformable_groups=2).cuda()
    torch.nn.init.constant_(m.weight, 1)
    off = torch.tensor([1, 1, 1, 0, 1, -1, 0, 1, 0, 0, 0, -1, -1, 1, -1, 0, -1, -1], dtype=torch.float32).cuda()
    off = off.unsqueeze(0).unsqueeze(-1).unsqueeze(-1).repeat(1, 2, 3, 3)
    x = torch.arange(18, dtype=torch.float32).view(1, 2, 3, 3).cuda()
    with torch.no_grad():
        pd = m(x, off)
    gt = torch.FloatTensor([81, 99, 117, 135, 153, 171, 189, 207, 225])
    assert (gt - pd.cpu().flatten()).abs().sum().item() < 1e-8


CASES = [
    # B, C, Co, H, W, k, stride, pad, dil, groups, dg, modulated
    (2, 64, 64, 20, 24, 3, 1, 1, 1, 1, 16, True),      # MVDualAttAlignment shape (arch.py:4242,3274)
    (1, 16, 16, 13, 17, 3, 1, 1, 1, 1, 16, True),      # DSTA shape (ops/attentionlayer.py:100)
    (2, 8, 12, 9, 11, 3, 2, 1, 1, 2, 4, True),
    (1, 8, 8, 10, 10, 3, 1, 2, 2, 1, 2, False),
    (1, 6, 300, 7, 9, 1, 1, 0, 1, 1, 3, True),         # > 256 output channels, 1x1 kernel
    (1, 4, 4, 40, 70, 3, 1, 1, 1, 1, 1, False),
]


@pytest.mark.parametrize("B,C,Co,H,W,k,s,p,d,g,dg,mod", CASES)
def test_dcn_forward_matches_oracle(B, C, Co, H, W, k, s, p, d, g, dg, mod):
    from cdfo_amd.dcn import deform_conv, modulated_deform_conv
    rs = np.random.RandomState(B * 1000 + C * 10 + Co)
    Ho = (H + 2 * p - (d * (k - 1) + 1)) // s + 1
    Wo = (W + 2 * p - (d * (k - 1) + 1)) // s + 1
    x = rs.standard_normal((B, C, H, W)).astype(np.float32)
    w = (rs.standard_normal((Co, C // g, k, k)) / np.sqrt(C // g * k * k)).astype(np.float32)
    b = rs.standard_normal((Co,)).astype(np.float32)
    off = (rs.standard_normal((B, 2 * dg * k * k, Ho, Wo)) * 3.0).astype(np.float32)   # some samples leave the image
    msk = rs.uniform(0, 1, (B, dg * k * k, Ho, Wo)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    with torch.no_grad():
        if mod:
            out = modulated_deform_conv(t(x), t(off), t(msk), t(w), t(b), s, p, d, g, dg)
            ref = dcn_forward_ref(x, off, msk, w, b, s, p, d, g, dg)
        else:
            out = deform_conv(t(x), t(off), t(w), s, p, d, g, dg)
            ref = dcn_forward_ref(x, off, None, w, None, s, p, d, g, dg)
    torch.cuda.synchronize()
    assert tuple(out.shape) == ref.shape
    err = np.abs(out.cpu().numpy() - ref).max()
    assert err <= 2e-5 * max(1.0, np.abs(ref).max()), err


def test_dcn_rejects_cpu_and_backward():
    from cdfo_amd.dcn import modulated_deform_conv
    x = torch.zeros(1, 4, 5, 5)
    with pytest.raises(NotImplementedError):
        modulated_deform_conv(x, torch.zeros(1, 18, 5, 5), torch.ones(1, 9, 5, 5), torch.zeros(4, 4, 3, 3), None, 1, 1, 1, 1, 1)
    xc = torch.zeros(1, 4, 5, 5, device="cuda", requires_grad=True)
    y = modulated_deform_conv(xc, torch.zeros(1, 18, 5, 5, device="cuda"), torch.ones(1, 9, 5, 5, device="cuda"),
                              torch.zeros(4, 4, 3, 3, device="cuda"), None, 1, 1, 1, 1, 1)
    with pytest.raises(NotImplementedError):
        y.sum().backward()
