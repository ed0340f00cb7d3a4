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


# --- SSIM: parity UNPINNED (cv2 absent, no reference-held vector).  What pins the restatement instead: hand-derived values.
C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2        # metric/psnr_ssim.py:329-330


def _hand_cases():
    """(a, b, crop, expected SSIM) derived WITHOUT any filtering code, from metric/psnr_ssim.py:320-351 on paper:
    * constant frames a = c1, b = c2: every window has mu = c, sigma = 0  =>  SSIM = (2 c1 c2 + C1) / (c1^2 + c2^2 + C1);
    * a vertical step edge a = 255 * [x >= e] against b = alpha * a: in a window whose Gaussian weight right of the edge is
      phi, mu1 = 255 phi, sigma1^2 = 255^2 phi (1 - phi), mu2 = alpha mu1, sigma2^2 = alpha^2 sigma1^2, sigma12 = alpha sigma1^2,
      with phi(x) = sum of the 1-D kernel's taps k with x + k >= e (cv2.getGaussianKernel(11, 1.5): exp(-(k-5)^2 / 4.5),
      normalised; rows do not matter), averaged over the valid columns [5, W-5)."""
    cases = [(np.full((30, 34), 100.0), np.full((30, 34), 120.0), 0, 0.9836109249983688),      # (24000 + C1) / (24400 + C1)
             (np.full((30, 34), 100.0), np.full((30, 34), 120.0), 4, 0.9836109249983688),
             (np.zeros((24, 28)), np.full((24, 28), 255.0), 0, 9.999000099990003e-05)]          # C1 / (255^2 + C1)
    taps = [np.exp(-((k - 5.0) ** 2) / 4.5) for k in range(11)]
    taps = [t / sum(taps) for t in taps]
    for (H, W, e, alpha, crop) in ((24, 40, 20, 0.5, 0), (32, 48, 25, 0.8, 4)):
        a = np.zeros((H, W))
        a[:, e:] = 255.0
        vals = []
        for x in range(crop, W - crop - 10):          # left column of a valid window in the un-cropped frame
            phi = sum(taps[k] for k in range(11) if x + k >= e)
            mu2, s = (255.0 * phi) ** 2, 255.0 ** 2 * phi * (1.0 - phi)
            vals.append((2 * alpha * mu2 + C1) * (2 * alpha * s + C2) / (((1 + alpha * alpha) * mu2 + C1) * ((1 + alpha * alpha) * s + C2)))
        cases.append((a, alpha * a, crop, float(np.mean(vals))))
    return cases


def test_oracle_ssim_hand_derived_vectors():
    """SSIM parity unpinned: cv2 absent.  The restatement must reproduce the closed-form values above."""
    from oracle.metrics_ref import calculate_ssim
    for a, b, crop, want in _hand_cases():
        assert abs(calculate_ssim(a, b, crop) - want) < 1e-12, (a.shape, crop, want)


def test_oracle_ssim_against_an_independent_filter():
    """Second opinion on the filtering itself: scipy.ndimage.correlate with the 11x11 outer-product window and OpenCV's default
    border (BORDER_REFLECT_101 = scipy 'mirror'), cropped [5:-5] exactly as metric/psnr_ssim.py:337-344 writes it."""
    from scipy import ndimage
    from oracle.metrics_ref import calculate_ssim, gaussian_kernel_11
    g = gaussian_kernel_11()
    win = np.outer(g, g)
    rs = np.random.RandomState(1)
    for shape in ((40, 48), (33, 57)):
        x = rs.randint(0, 256, shape).astype(np.float64)
        y = np.clip(x + rs.randn(*shape) * 12, 0, 255)
        f = lambda im: ndimage.correlate(im, win, mode="mirror")[5:-5, 5:-5]      # noqa: E731
        mu1, mu2 = f(x), f(y)
        s11, s22, s12 = f(x * x) - mu1 ** 2, f(y * y) - mu2 ** 2, f(x * y) - mu1 * mu2
        want = (((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 ** 2 + mu2 ** 2 + C1) * (s11 + s22 + C2))).mean()
        assert abs(calculate_ssim(x, y, 0) - want) < 1e-12


@pytest.mark.gpu
def test_device_ssim_hand_derived_vectors():
    """The device kernel on the same closed-form cases (SSIM parity unpinned: cv2 absent)."""
    from cdfo_amd import metrics as M
    for a, b, crop, want in _hand_cases():
        got = M.calculate_ssim(torch.from_numpy(a).float().cuda()[None], torch.from_numpy(b).float().cuda()[None], crop,
                               from_unit_range=False).item()
        assert abs(got - want) < 1e-9, (a.shape, crop, got, want)


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
