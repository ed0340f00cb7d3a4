"""ORACLE (test infrastructure only): numpy fp64 restatement of the reference's PSNR / SSIM
(metric/psnr_ssim.py:278-317 calculate_psnr, :320-351 _ssim, :353-399 calculate_ssim) for single-channel frames.

calculate_psnr is pinned by tests/golden/metrics_psnr.npz (vectors of the reference's own function).  _ssim calls
cv2.getGaussianKernel / cv2.filter2D; cv2 (opencv-python, unpinned in the reference) is absent from this image, so for
SSIM parity is UNPINNED: the restatement follows the published algorithm (Wang et al. 2004; OpenCV's documented kernel
G_i = alpha * exp(-(i-5)^2 / (2*1.5^2)), sum 1; filter2D = correlation, and the [5:-5] crop keeps only positions whose
window lies inside the image, so the border mode never matters)."""
from __future__ import annotations

import numpy as np


def calculate_psnr(img1: np.ndarray, img2: np.ndarray, crop_border: int) -> float:
    a, b = img1.astype(np.float64), img2.astype(np.float64)
    if crop_border != 0:
        a = a[crop_border:-crop_border, crop_border:-crop_border, ...]
        b = b[crop_border:-crop_border, crop_border:-crop_border, ...]
    mse = np.mean((a - b) ** 2)
    return float("inf") if mse == 0 else 20.0 * np.log10(255.0 / np.sqrt(mse))


def gaussian_kernel_11() -> np.ndarray:
    g = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5 ** 2))
    return g / g.sum()


def _filter_valid(img: np.ndarray, g: np.ndarray) -> np.ndarray:
    H, W = img.shape
    tmp = np.zeros((H - 10, W), np.float64)
    for k in range(11):
        tmp += g[k] * img[k:k + H - 10, :]
    out = np.zeros((H - 10, W - 10), np.float64)
    for k in range(11):
        out += g[k] * tmp[:, k:k + W - 10]
    return out


def calculate_ssim(img1: np.ndarray, img2: np.ndarray, crop_border: int) -> float:
    a, b = img1.astype(np.float64), img2.astype(np.float64)
    if crop_border != 0:
        a = a[crop_border:-crop_border, crop_border:-crop_border]
        b = b[crop_border:-crop_border, crop_border:-crop_border]
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    g = gaussian_kernel_11()
    mu1, mu2 = _filter_valid(a, g), _filter_valid(b, g)
    s11 = _filter_valid(a * a, g) - mu1 ** 2
    s22 = _filter_valid(b * b, g) - mu2 ** 2
    s12 = _filter_valid(a * b, g) - mu1 * mu2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 ** 2 + mu2 ** 2 + C1) * (s11 + s22 + C2))
    return float(m.mean())
