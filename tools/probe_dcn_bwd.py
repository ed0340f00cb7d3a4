import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from cdfo_amd import deform_conv_cuda as ext
from cdfo_amd.dcn import modulated_deform_conv
from oracle.dcn_modules_ref import dcn_backward_ref
B, C, Co, H, W, k, dg = 1, 64, 64, 24, 40, 3, 16
rs = np.random.RandomState(1)
x = rs.standard_normal((B, C, H, W)).astype(np.float32)
w = (rs.standard_normal((Co, C, k, k)) / 24).astype(np.float32)
b = rs.standard_normal((Co,)).astype(np.float32)
off = (rs.standard_normal((B, 2 * dg * 9, H, W)) * 3).astype(np.float32)
msk = rs.uniform(0, 1, (B, dg * 9, H, W)).astype(np.float32)
go = rs.standard_normal((B, Co, H, W)).astype(np.float32)
ref = dcn_backward_ref(x, off, msk, w, go, 1, 1, 1, 1, dg)
t = lambda a: torch.from_numpy(a).cuda()
e = torch.empty(0, device="cuda")
for fill in (0.0, 7.0):
    gi, gw, gb = torch.zeros_like(t(x)), torch.zeros_like(t(w)), torch.zeros_like(t(b))
    goff, gm = torch.full_like(t(off), fill), torch.full_like(t(msk), fill)
    ext.modulated_deform_conv_cuda_backward(t(x), t(w), t(b), e, t(off), t(msk), e, gi, gw, gb, goff, gm, t(go), k, k, 1, 1, 1, 1, 1, 1, 1, dg, True)
    torch.cuda.synchronize()
    for name, got in (("grad_input", gi), ("grad_offset", goff), ("grad_mask", gm), ("grad_weight", gw), ("grad_bias", gb)):
        r = ref[name]
        print(f"prefill {fill}: {name} rel err {np.abs(got.cpu().numpy() - r).max() / np.abs(r).max():.2e}")
# through the Function with requires_grad on a subset
tx, toff, tm, tw, tb = (t(a).requires_grad_() for a in (x, off, msk, w, b))
junk = torch.full((64 << 20,), 3.0, device="cuda"); del junk      # leave non-zero bytes in the caching allocator
out = modulated_deform_conv(tx, toff, tm, tw, tb, 1, 1, 1, 1, dg)
out.backward(t(go))
for name, got in (("grad_input", tx.grad), ("grad_offset", toff.grad), ("grad_mask", tm.grad), ("grad_weight", tw.grad), ("grad_bias", tb.grad)):
    r = ref[name]
    print(f"Function: {name} rel err {np.abs(got.cpu().numpy() - r).max() / np.abs(r).max():.2e}")
