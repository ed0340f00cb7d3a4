"""Developer probe (GPU box only): s_memtime stamps inside one steady-state batch of the Winograd kernel (dbg 512), per wave."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdfo_amd import kernels as K

B, H, W, s2d = 8, 544, 960, True
UP = "--up" in sys.argv            # the UP form: the same launch from the low-resolution source (bilinear x2 inside the input transform)
argv = [a for a in sys.argv[1:] if a != "--up"]
pc = K.pack_conv(torch.randn(256, 64, 3, 3, device="cuda") / 24.0, torch.randn(256, device="cuda"))
clk = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")
DBG = int(argv[0]) if argv else 512
if UP:
    from cdfo_amd import _lib
    src = K.to_cp16(torch.randn(B, H // 2, W // 2, 64, device="cuda"))
    out = K.conv3x3_wino_up2(src, pc, act=1)
    for _ in range(3):
        K.check(_lib.lib().cdfo_conv3x3_c64_wino_dbg(K._vp(src), B, H, W, K._vp(pc.ww), K._vp(pc.bias), pc.Cout, 1, K._vp(out), 2, -2,
                                                     K._vp(clk), K._stream()), "cdfo_conv3x3_c64_wino_dbg")
else:
    src = K.to_cp16(torch.randn(B, H, W, 64, device="cuda"))
    out = K.conv3x3_wino(src, pc, act=1, s2d=s2d)
    for _ in range(3):
        K.conv3x3_wino(src, pc, act=1, s2d=s2d, out=out, dbg=DBG, clk=clk)
torch.cuda.synchronize()
c = clk.view(256, 8, 8).cpu().double()
ok = c[:, :, 7] > 0
names = ["entry", "row0 MFMAs issued", "row0 epilogue", "mid (V of next batch, loads)", "row1 + epilogue", "barrier passed", "row2 MFMAs issued", "end (row2 epilogue)"]
print(f"workgroups with stamps: {int(ok[:, 0].sum())} / 256; batch length (entry -> end), cycles: median {(c[:, :, 7] - c[:, :, 0])[ok].median():.0f}")
for grp, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7 (deferred last epilogue)", slice(4, 8))):
    d = (c[:, sl, 1:] - c[:, sl, :-1])[ok[:, sl]]
    print(grp + ": median cycles per segment")
    for k in range(7):
        print(f"   {names[k]:32s} -> {names[k + 1]:32s} {d[:, k].median():8.0f}   (p10 {d[:, k].quantile(0.1):.0f}, p90 {d[:, k].quantile(0.9):.0f})")
# skew between the SIMD partners at batch entry
sk = (c[:, 4:8, 0] - c[:, 0:4, 0])[ok[:, 0:4] & ok[:, 4:8]]
print(f"entry skew waves 4-7 minus waves 0-3: median {sk.median():.0f} cycles (p10 {sk.quantile(0.1):.0f}, p90 {sk.quantile(0.9):.0f})")
