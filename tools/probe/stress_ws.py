import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cdfo_amd import kernels as K
B, H, W, Cout = 8, 544, 960, 256
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(B, H, W, 64, device="cuda", generator=g)
w = torch.randn(Cout, 64, 3, 3, device="cuda", generator=g) / 24.0
b = torch.randn(Cout, device="cuda", generator=g)
pc = K.pack_conv(w, b)
ref = K.conv([x.half()], pc, pad=1, act=1, out_f16=True)
src = K.to_cp16(x)
out = K.conv3x3_ws(src, pc, act=1)
for dbg in [int(a) for a in sys.argv[1].split(",")]:
    tot = 0
    runs = int(sys.argv[2])
    detail = []
    for it in range(runs):
        out.zero_()
        K.conv3x3_ws(src, pc, act=1, out=out, dbg=dbg)
        bad = ((K.from_cp16(out).float() - ref.float()).abs() > 0.03)
        n = int(bad.sum().item())
        tot += n
        if n and len(detail) < 6:
            idx = bad.nonzero()
            rows = torch.unique(idx[:, 1]).tolist()
            xs = (int(idx[:, 2].min()), int(idx[:, 2].max()))
            cs = (int(idx[:, 3].min()), int(idx[:, 3].max()))
            detail.append((n, rows[:6], xs, cs))
    print(f"dbg {dbg}: {runs} runs, bad elements total {tot}; {detail}", flush=True)
