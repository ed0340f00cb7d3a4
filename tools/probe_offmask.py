import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdfo_amd import nchw_autograd as G
torch.manual_seed(0)
B, H, W, dg = 1, 24, 40, 16
t = 9 * dg
h1 = torch.randn(B, H, W, 3 * t, device="cuda", requires_grad=True)
h2 = torch.randn(B, H, W, 3 * t, device="cuda", requires_grad=True)
flow = torch.randn(B, 2, H, W, device="cuda")
go, gm = torch.randn(B, 2 * t, H, W, device="cuda"), torch.randn(B, t, H, W, device="cuda")
off, msk = G.offset_mask(h1, h2, flow, t, 10.0)
((off * go).sum() + (msk * gm).sum()).backward()
a1, a2 = h1.detach().double().requires_grad_(True), h2.detach().double().requires_grad_(True)
n1, n2 = a1.permute(0, 3, 1, 2), a2.permute(0, 3, 1, 2)
roff = 10.0 * torch.tanh(n1[:, :2 * t]) + 10.0 * torch.tanh(n2[:, :2 * t]) + flow.double().flip(1).repeat(1, t, 1, 1)
rmsk = torch.sigmoid(n1[:, 2 * t:] + n2[:, 2 * t:])
((roff * go.double()).sum() + (rmsk * gm.double()).sum()).backward()
rel = lambda a, r: ((a.double() - r).abs().max() / r.abs().max()).item()
print("fwd off", rel(off, roff), "mask", rel(msk, rmsk), "g1", rel(h1.grad, a1.grad), "g2", rel(h2.grad, a2.grad))
