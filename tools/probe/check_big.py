import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from cdfo_amd import kernels as K
B, H, W, Cout = 8, 544, 960, 256
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(B, H, W, 64, device="cuda", generator=g)
w = torch.randn(Cout, 64, 3, 3, device="cuda", generator=g) / 24.0
b = torch.randn(Cout, device="cuda", generator=g)
pc = K.pack_conv(w, b)
x16 = x.half()
for s2d in (True, False):
    t = K.conv([x16], pc, pad=1, act=1, s2d=s2d, out_f16=True)
    o = K.conv3x3_ws(K.to_cp16(x), pc, act=1, s2d=s2d)
    torch.cuda.synchronize()
    for b0 in range(B):
        ref = F.leaky_relu(F.conv2d(x16[b0:b0+1].float().permute(0, 3, 1, 2), w.half().float(), b, padding=1), 0.1)[0]  # [C,H,W]
        def unpack(v):
            v = v[b0].float()
            if s2d:
                v = v.view(H // 2, W // 2, 2, 2, Cout).permute(0, 2, 1, 3, 4).reshape(H, W, Cout)
            return v.permute(2, 0, 1)
        et = (unpack(t) - ref).abs()
        eo = (unpack(o) - ref).abs()
        print(f"s2d={s2d} img {b0}: tiled err {et.max().item():.3e}  ws err {eo.max().item():.3e}", flush=True)
        if et.max() > 0.05:
            idx = (et > 0.05).nonzero()
            print("   tiled bad count", idx.shape[0], "first", idx[:3].tolist(), "last", idx[-3:].tolist())
        if eo.max() > 0.05:
            idx = (eo > 0.05).nonzero()
            print("   ws bad count", idx.shape[0], "first", idx[:3].tolist(), "last", idx[-3:].tolist())
