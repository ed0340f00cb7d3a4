import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from cdfo_amd import autograd as A, kernels as K
torch.manual_seed(0)
for prec in (K.PREC_F32, K.PREC_BF16X3):
    A.CONV_PREC = prec
    for Ci, Co in ((64, 432), (64, 64), (64, 128), (432, 64), (64, 48)):
        x = torch.randn(1, 24, 40, Ci, device="cuda", requires_grad=True)
        w = (torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05).requires_grad_(True)
        b = torch.randn(Co, device="cuda", requires_grad=True)
        g = torch.randn(1, 24, 40, Co, device="cuda")
        y = A.conv(x, w, b, 1, 1)
        (y * g).sum().backward()
        xr = x.detach().double().permute(0, 3, 1, 2).requires_grad_(True)
        wr, br = w.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
        yr = F.conv2d(xr, wr, br, padding=1)
        (yr * g.double().permute(0, 3, 1, 2)).sum().backward()
        rel = lambda a, r: ((a.double() - r).abs().max() / r.abs().max()).item()
        print(f"prec {prec} {Ci}->{Co}: y {rel(y.permute(0,3,1,2), yr):.2e} dx {rel(x.grad.permute(0,3,1,2), xr.grad):.2e} dW {rel(w.grad, wr.grad):.2e} db {rel(b.grad, br.grad):.2e}")
