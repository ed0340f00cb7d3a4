import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_oracle_grad_golden import oracle_grads
from test_gpu_train import _hip_step
from oracle.cvsr_v8_ref import make_inputs
B, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
fake = dict(B=B, H=H, W=W, wseed=22, iseed=202, hr_seed=209, stride=53)
_, _, g64 = oracle_grads(fake, torch.float64)
_, _, g32 = oracle_grads(fake, torch.float32)
inp = make_inputs(B, H, W, 202, "b1n")
hr = torch.from_numpy(np.random.RandomState(209).uniform(0, 1, (B, 1, 4 * H, 4 * W)).astype(np.float32))
m, out, loss, grads = _hip_step(22, inp, hr, [u.cuda() for u in inp["gumbel_u"]])
rows = []
for k, go in g64.items():
    if go is None: continue
    s = go.abs().max().item()
    if s == 0: continue
    e = (grads[k].cpu().double() - go).abs().max().item() / s
    e32 = (g32[k].double() - go).abs().max().item() / s
    rows.append((e, e32, s, k))
rows.sort(reverse=True)
print(f"B={B} {H}x{W}: HIP-vs-f64 / cpu32-vs-f64 / max|g|")
for r in rows[:14]: print("  %.2e  %.2e  %.2e  %s" % r)
print("  median HIP %.2e  median cpu32 %.2e" % (np.median([r[0] for r in rows]), np.median([r[1] for r in rows])))
