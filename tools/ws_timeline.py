"""Developer tool (GPU box only): s_memtime stamps inside the ring-fed weights-stationary kernel (conv3x3_ws.hip, dbg 32).
    python tools/ws_timeline.py [H W]      (default 544 960, s2d store)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdfo_amd import kernels as K

NAMES = ["entry", "barrier", "tap2", "tap5", "tap8", "end"]


def main():
    H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (544, 960)
    B, Cout = 8, 256
    x = torch.randn(B, H, W, 64, device="cuda")
    pc = K.pack_conv(torch.randn(Cout, 64, 3, 3, device="cuda") / 24.0, torch.randn(Cout, device="cuda"))
    src = K.to_cp16(x)
    clk = torch.zeros(256, 12, 4, 8, dtype=torch.int64, device="cuda")
    s2d = H == 544
    out = K.conv3x3_ws(src, pc, act=1, s2d=s2d)
    for _ in range(3):
        K.conv3x3_ws(src, pc, act=1, s2d=s2d, out=out, dbg=32, clk=clk)
    torch.cuda.synchronize()
    t = clk.cpu().double()[:, :8]                  # consumers
    nst = 4 if (t[:, :, 3, 0] > 0).any() else 2    # steps per tile: 4 chunks (32x32x16 form) or 2 superchunks (16x16x32 form)
    ok = (t[:, :, :nst, 0] > 0).all(dim=-1).all(dim=-1)
    t = t[ok]
    print(f"# 64->256 {H}x{W} B{B}: {int(ok.sum())} workgroups; ticks = shader cycles")
    d = t[:, :, :nst, 1:6] - t[:, :, :nst, 0:5]
    for i in range(5):
        print(f"   {NAMES[i]:8s} -> {NAMES[i + 1]:8s} {d[..., i].mean():7.1f}   (group A {d[:, :4, :, i].mean():7.1f}, group B {d[:, 4:, :, i].mean():7.1f})")
    period = (t[:, :, 1, 6] - t[:, :, 0, 6])
    print(f"# undisturbed tile period (epilogue start of tile 4 -> tile 5): {period[period > 0].mean():.0f} ticks")
    long = (t[:, :, 2, 6] - t[:, :, 1, 6])
    if (long > 0).any():
        print(f"# mean tile period over tiles 5..25: {long[long > 0].mean() / 20:.0f} ticks")
    g = t[0]
    base = g[:, 0, 0].min()
    for wv in range(8):
        for c in range(nst):
            print(f"   wave {wv} step {c}: " + " ".join(f"{int(v - base):6d}" for v in g[wv, c, :6]))


if __name__ == "__main__":
    main()
