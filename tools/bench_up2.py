"""Developer micro-benchmark (GPU box only): Block_'s x2 branch source -- materialised (block_prologue's u16 + conv3x3_wino s2d) against
the on-the-fly form (block_prologue lowres_up + conv3x3_wino_up2)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdfo_amd import kernels as K
from bench_conv import timeit

for (B, H, W) in [(8, 272, 480), (1, 544, 960), (4, 120, 240)]:
    x = torch.randn(B, H, W, 64, device="cuda")
    w = torch.randn(256, 64, 3, 3, device="cuda") / 24.0
    pc = K.pack_conv(w, torch.randn(256, device="cuda"))
    wu, bu, wd, bd = (torch.randn(64, 64, 1, 1, device="cuda") / 8, torch.randn(64, device="cuda"), torch.randn(64, 64, 1, 1, device="cuda") / 8,
                      torch.randn(64, device="cuda"))
    pro = K.pack_block_prologue(wu, bu, wd, bd)
    u16, d16 = K.block_prologue(x, pro)
    t16, _ = K.block_prologue(x, pro, lowres_up=True)
    a = K.conv3x3_wino(u16, pc, act=1, s2d=True)
    b = K.conv3x3_wino_up2(t16, pc, act=1)
    d = (a.float() - b.float()).abs().max().item()
    ms = [timeit(lambda: K.block_prologue(x, pro)), timeit(lambda: K.conv3x3_wino(u16, pc, act=1, s2d=True)),
          timeit(lambda: K.block_prologue(x, pro, lowres_up=True)), timeit(lambda: K.conv3x3_wino_up2(t16, pc, act=1))]
    print(f"B{B} {H}x{W}: materialised prologue {ms[0]:.3f} + conv {ms[1]:.3f} = {ms[0] + ms[1]:.3f} ms | on the fly prologue {ms[2]:.3f} + conv {ms[3]:.3f} "
          f"= {ms[2] + ms[3]:.3f} ms | max diff of the two results {d:.2e} (scale {a.float().abs().max().item():.2f})", flush=True)
