"""Developer probe (GPU box; lives under tests/ because it uses the CPU oracle): clip 0 of the c3 batch against the oracle for a list of
arithmetic switches of CVSR_V8, with the step time of each.  usage: python tests/probe_precision.py [B H W]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arch.SIDECVSR_our import CVSR_V8  # noqa: E402
from oracle.cvsr_v8_ref import cvsr_v8_forward, make_inputs, make_state_dict  # noqa: E402

B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (8, 272, 480)
sd = make_state_dict(0, perturb=True)
inp = make_inputs(B, H, W, 1002, pad_rows={272: 2, 544: 4}.get(H, 0))
t0 = time.time()
with torch.no_grad():
    ref, L1_ref = cvsr_v8_forward(sd, inp["x"][:1], None, inp["mvs1"][:1], inp["pms"][:1], inp["rms"][:1], inp["ufs"][:1], None,
                                  [u[:1] for u in inp["gumbel_u"]])
print(f"oracle clip 0: {time.time() - t0:.1f} s", flush=True)
d = {k: v.cuda() for k, v in inp.items() if k != "gumbel_u"}
noise = [u.cuda() for u in inp["gumbel_u"]]

CONFIGS = [("default", {}), ("fe_weight_lo=False", {"fe_weight_lo": False})]
for name, attrs in CONFIGS:
    m = CVSR_V8()
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    for k, v in attrs.items():
        setattr(m, k, v)
    with torch.no_grad():
        out, L1 = m(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"], gumbel_uniform=noise)
        torch.cuda.synchronize()
        e_out = (out[:1].cpu() - ref).abs().max().item()
        e_l1 = (L1[:7].cpu() - L1_ref).abs().max().item()
        for _ in range(2):
            m(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"], gumbel_uniform=noise)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 8
        for _ in range(n):
            m(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"], gumbel_uniform=noise)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
    print(f"{name:28s} out {e_out:.2e}  L1_fea {e_l1:.2e}  {ms:.2f} ms/step  {m.last_range}", flush=True)
    del m
    torch.cuda.empty_cache()
