"""Developer probe (GPU box): inside a real forward, compare every conv3x3_wino call with the direct kernel on the same operands."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdfo_amd import kernels as K
from arch.SIDECVSR_our import CVSR_V8
from _inputs import random_inputs

B, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(0)
m = CVSR_V8().cuda().eval()
d = random_inputs(B, H, W)
orig = K.conv3x3_wino
stats = []


def checked(src, pc, **kw):
    o = orig(src, pc, **kw)
    torch.cuda.synchronize()
    r = K.conv3x3_ws(src, pc, **kw)
    torch.cuda.synchronize()
    diff = (o.float() - r.float()).abs()
    e = diff.max().item()
    bad = (diff > 0.05).nonzero()
    stats.append((tuple(src.shape), kw.get("s2d", False), e, r.float().abs().max().item(), bad.shape[0], bad[:4].tolist(), bad[-2:].tolist()))
    return o


K.conv3x3_wino = checked
m.trunk_side_stream = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
with torch.no_grad():
    m(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"], gumbel_uniform=d["gumbel_u"])
torch.cuda.synchronize()
for s in stats[:12]:
    print(s)
print("worst", max(stats, key=lambda s: s[2]))
