"""Developer probe (GPU box): where one wino unit case differs from the direct kernel (image, 16-channel block, row, strip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdfo_amd import kernels as K

Cout, H, W, B, s2d, act = [int(v) for v in sys.argv[1:7]] if len(sys.argv) > 6 else (384, 20, 70, 2, 1, 1)
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 5
g = torch.Generator().manual_seed(Cout + H + W)
x = torch.randn(B, 64, H, W, generator=g)
w = torch.randn(Cout, 64, 3, 3, generator=g) / 24.0
b = torch.randn(Cout, generator=g)
pc = K.pack_conv(w.cuda(), b.cuda())
src = K.to_cp16(x.permute(0, 2, 3, 1).contiguous().cuda())

def nchw(t):
    t = K.from_cp16(t).float().cpu()
    if s2d:
        t = t.view(B, H // 2, W // 2, 2, 2, Cout).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, Cout)
    return t.permute(0, 3, 1, 2)

direct = nchw(K.conv3x3_ws(src, pc, act=act, s2d=bool(s2d)))
for r in range(reps):
    got = nchw(K.conv3x3_wino(src, pc, act=act, s2d=bool(s2d)))
    bad = (got - direct).abs() > 0.05
    print(f"rep {r}: {int(bad.sum())} bad values, max {float((got - direct).abs().max()):.3f}")
    if bad.any():
        idx = bad.nonzero()
        keys = {}
        for bi, c, y, xx in idx.tolist():
            keys.setdefault((bi, c // 16, y, xx // 32), 0)
            keys[(bi, c // 16, y, xx // 32)] += 1
        for k in sorted(keys)[:60]:
            print("   image %d cb %2d row %2d strip %d: %d" % (*k, keys[k]))
            bi, cbk, y, st = k
            sub = bad[bi, cbk * 16:cbk * 16 + 16, y, st * 32:st * 32 + 32]
            print("      columns:", sorted(set(sub.nonzero()[:, 1].tolist())), "channels:", sorted(set(sub.nonzero()[:, 0].tolist())))
            dd = (got - direct)[bi, cbk * 16:cbk * 16 + 16, y, st * 32:st * 32 + 32]
            cols = sorted(set(sub.nonzero()[:, 1].tolist()))[:2]
            for cx in cols:
                print("      diff col", cx, [round(v, 2) for v in dd[:, cx].tolist()])
