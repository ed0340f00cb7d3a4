import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arch.SIDECVSR_our import CVSR_V8
from oracle.cvsr_v8_ref import make_inputs, make_state_dict
sd = make_state_dict(3)
inp = make_inputs(1, 16, 24, 78)
bad = {k: v.clone() for k, v in sd.items()}
p = "recon_trunk.body.2.body.1.body."
for e in (20, 24):
    b = {k: v.clone() for k, v in sd.items()}
    b[p + "0.weight"] *= 2.0 ** e; b[p + "0.bias"] *= 2.0 ** e; b[p + "2.weight"] *= 2.0 ** -e
    m = CVSR_V8(); m.load_state_dict(b, strict=True); m = m.cuda().eval()
    m.debug_taps = {}
    d = {k: v.cuda() for k, v in inp.items() if k != "gumbel_u"}
    noise = [u.cuda() for u in inp["gumbel_u"]]
    with torch.no_grad():
        out, L1 = m(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"], gumbel_uniform=noise)
    torch.cuda.synchronize()
    pend = m.__dict__.get("_pending_guard")
    print("scale 2^%d" % e, "out nonfinite", int((~torch.isfinite(out)).sum()), "trunk nonfinite", int((~torch.isfinite(m.debug_taps["trunk"])).sum()),
          "pending", pend is not None, "host", None if pend is None else pend[1].tolist(), "probe", None if pend is None else pend[5].tolist(),
          "w absmax", b[p + "0.weight"].abs().max().item(), m.__dict__.get("_last_range"))
