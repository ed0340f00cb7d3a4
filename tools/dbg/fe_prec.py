import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, time
from arch.SIDECVSR_our import CVSR_V8
from oracle.cvsr_v8_ref import cvsr_v8_forward, make_inputs, make_state_dict
sd = make_state_dict(14)
inp = make_inputs(1, 24, 40, 105)
with torch.no_grad():
    ref, L1r = cvsr_v8_forward(sd, inp["x"], None, inp["mvs1"], inp["pms"], inp["rms"], inp["ufs"], None, inp["gumbel_u"])
for wlo in (True, False):
    m = CVSR_V8(); m.load_state_dict(sd, strict=True); m = m.cuda().eval(); m.fe_weight_lo = wlo
    d = {k: v.cuda() for k, v in inp.items() if k != "gumbel_u"}
    with torch.no_grad():
        out, L1 = m(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"], gumbel_uniform=[u.cuda() for u in inp["gumbel_u"]])
    print(f"fe_weight_lo={wlo}: out {(out.cpu()-ref).abs().max().item():.2e}  L1_fea {(L1.cpu()-L1r).abs().max().item():.2e}  (max |L1| {L1r.abs().max().item():.2f})")
