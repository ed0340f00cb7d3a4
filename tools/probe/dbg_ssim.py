import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cdfo_amd import metrics as M
from oracle.metrics_ref import calculate_psnr, calculate_ssim
g = np.load("tests/golden/metrics_psnr.npz")
a, b = torch.from_numpy(g["a"][..., 0].astype(np.float32)), torch.from_numpy(g["b"][..., 0].astype(np.float32))
for crop in (0, 4):
    s = M.calculate_ssim(a.cuda(), b.cuda(), crop, from_unit_range=False).cpu().numpy()
    p = M.calculate_psnr(a.cuda(), b.cuda(), crop, from_unit_range=False).cpu().numpy()
    for n in range(4):
        print(crop, n, s[n], calculate_ssim(g["a"][n, :, :, 0], g["b"][n, :, :, 0], crop), p[n], calculate_psnr(g["a"][n, :, :, 0], g["b"][n, :, :, 0], crop))
