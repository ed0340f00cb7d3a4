"""Developer tool: run ONE conv shape a few times (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdfo_amd import kernels as K
Cin, Cout, H, W, B, prec = [int(v) for v in sys.argv[1:7]]
x = torch.randn(B, H, W, Cin, device="cuda")
w = torch.randn(Cout, Cin, 3, 3, device="cuda") / (Cin * 9) ** 0.5
pc = K.pack_conv(w, torch.randn(Cout, device="cuda"))
out = K.empty_act(B, H, W, Cout, x.device)
for _ in range(3):
    K.conv([x], pc, pad=1, act=1, out=out, prec=prec)
torch.cuda.synchronize()
