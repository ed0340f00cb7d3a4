"""Developer probe (GPU box): CVSR_V7 at the benchmark's frame size in both 16-bit modes -- where do non-finite values appear?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arch.SIDECVSR_our import CVSR_V7  # noqa: E402
from oracle.cvsr_v7_ref import make_inputs_v7, make_state_dict_v7  # noqa: E402

B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (3, 272, 480)
m = CVSR_V7()
m.load_state_dict(make_state_dict_v7(0), strict=True)
m = m.cuda().eval()
inp = make_inputs_v7(B, H, W, 2001)
dev = {k: v.cuda() for k, v in inp.items() if k != "gumbel_u"}
noise = [u.cuda() for u in inp["gumbel_u"]]
res = {}
for prec in ("bf16x3", "fp16x2"):
    m.precision = prec
    m.debug_taps = {}
    with torch.no_grad():
        out, L1 = m(dev["x"], dev["mvs0"], dev["mvs1"], dev["pms"], dev["rms"], dev["ufs"], gumbel_uniform=noise)
    torch.cuda.synchronize()
    taps = {k: v for k, v in m.debug_taps.items() if torch.is_tensor(v)}
    res[prec] = (out, L1, taps)
    bad = {k: int((~torch.isfinite(v)).sum().item()) for k, v in taps.items()}
    print(prec, "out nonfinite", int((~torch.isfinite(out)).sum().item()), "absmax", out[torch.isfinite(out)].abs().max().item(),
          "| taps nonfinite:", {k: n for k, n in bad.items() if n}, "| tap absmax:",
          {k: round(v[torch.isfinite(v)].abs().max().item(), 3) for k, v in taps.items()}, flush=True)
a, b = res["bf16x3"], res["fp16x2"]
print("out diff", (a[0] - b[0]).abs().max().item(), "L1 diff", (a[1] - b[1]).abs().max().item())
for k in a[2]:
    if k in b[2] and a[2][k].shape == b[2][k].shape:
        print(" tap", k, tuple(a[2][k].shape), (a[2][k] - b[2][k]).abs().max().item())
if not torch.isfinite(b[0]).all():
    idx = (~torch.isfinite(b[0])).nonzero()
    print("first / last bad out index", idx[0].tolist(), idx[-1].tolist(), "count", idx.shape[0])
