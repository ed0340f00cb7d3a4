import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from cdfo_amd import deform_conv_cuda as ext
from cdfo_amd.dcn import modulated_deform_conv
from oracle.dcn_modules_ref import dcn_forward_ref
B, C, Co, H, W, dg = 8, 64, 64, 272, 480, 16
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, C, H, W, device="cuda", generator=g)
w = torch.randn(Co, C, 3, 3, device="cuda", generator=g) / 24
b = torch.randn(Co, device="cuda", generator=g)
off = 2 * torch.randn(B, 2 * dg * 9, H, W, device="cuda", generator=g)
msk = torch.rand(B, dg * 9, H, W, device="cuda", generator=g)
if os.environ.get("WITH_ORACLE"):
    n = lambda t: t[:1].cpu().numpy()
    ref = torch.from_numpy(dcn_forward_ref(n(x), n(off), n(msk), w.cpu().numpy(), b.cpu().numpy(), 1, 1, 1, 1, dg)).cuda()
with torch.no_grad():
    ext.EXACT_FP32 = True
    oe = modulated_deform_conv(x, off, msk, w, b, 1, 1, 1, 1, dg)
    ext.EXACT_FP32 = False
    res = []
    for i in range(3):
        o = modulated_deform_conv(x, off, msk, w, b, 1, 1, 1, 1, dg)
        res.append(round((o - oe).abs().max().item(), 7))
    msg = f"DBG={os.environ.get('CDFO_DCN_DBG','0')}: fast vs exact {res}"
    if os.environ.get("WITH_ORACLE"):
        msg += f"  | vs C oracle (image 0): exact {(oe[:1] - ref).abs().max().item():.2e} fast {(o[:1] - ref).abs().max().item():.2e}"
    print(msg)
