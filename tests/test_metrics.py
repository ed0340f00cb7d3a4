"""PSNR / SSIM: oracle vs vectors of the reference's calculate_psnr (CPU), device kernels vs the
oracle (GPU)."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden", "metrics_psnr.npz")


def test_oracle_psnr_matches_reference_vectors():
    from oracle.metrics_ref import calculate_psnr
    g = np.load(GOLD)
    for x, y, row in zip(g["a"], g["b"], g["psnr"]):
        for c, want in zip(g["crops"], row):
            got = calculate_psnr(x.astype(np.float64), y.astype(np.float64), int(c))
            assert got == want or abs(got - want) < 1e-12


def test_oracle_ssim_properties():
    from oracle.metrics_ref import calculate_ssim, gaussian_kernel_11
    assert abs(gaussian_kernel_11().sum() - 1.0) < 1e-15
    rs = np.random.RandomState(0)
    a = rs.randint(0, 256, size=(40, 48)).astype(np.float64)
    assert abs(calculate_ssim(a, a, 4) - 1.0) < 1e-12                        # identical frames
    b = np.clip(a + rs.randn(40, 48) * 10, 0, 255)
    assert abs(calculate_ssim(a, b, 4) - calculate_ssim(b, a, 4)) < 1e-12    # symmetric
    assert calculate_ssim(a, b, 4) < 1.0


@pytest.mark.gpu
def test_device_metrics_match_oracle():
    from cdfo_amd import metrics as M
    from oracle.metrics_ref import calculate_psnr, calculate_ssim
    g = np.load(GOLD)
    a, b = torch.from_numpy(g["a"][..., 0].astype(np.float32)), torch.from_numpy(g["b"][..., 0].astype(np.float32))
    for crop in (0, 4):
        p = M.calculate_psnr(a.cuda(), b.cuda(), crop, from_unit_range=False).cpu().numpy()
        s = M.calculate_ssim(a.cuda(), b.cuda(), crop, from_unit_range=False).cpu().numpy()
        for n in range(a.shape[0]):
            want_p = calculate_psnr(g["a"][n, :, :, 0], g["b"][n, :, :, 0], crop)
            assert (np.isinf(want_p) and np.isinf(p[n])) or abs(p[n] - want_p) < 1e-9
            assert abs(s[n] - calculate_ssim(g["a"][n, :, :, 0], g["b"][n, :, :, 0], crop)) < 1e-9
    # network-output form: [0,1] values, clamp + x255 (+ PNG rounding)
    x = torch.rand(2, 1, 64, 80) * 1.2 - 0.1
    y = (x + torch.randn_like(x) * 0.02)
    for r8 in (False, True):
        def conv(t, r8=r8):
            v = np.clip(t.numpy().astype(np.float32) * np.float32(255.0), 0, 255)
            return np.rint(v) if r8 else v
        p = M.calculate_psnr(x.cuda(), y.cuda(), 4, round8=r8).cpu().numpy()
        for n in range(2):
            assert abs(p[n] - calculate_psnr(conv(x[n, 0]), conv(y[n, 0]), 4)) < 1e-6
