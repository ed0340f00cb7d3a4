"""Developer tool (GPU box only): wall time of the forward's phases at c3 (events on the caller's stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arch.SIDECVSR_our import CVSR_V8
from _inputs import random_inputs


def main():
    B, H, W = 8, 272, 480
    m = CVSR_V8()
    m = m.cuda().eval()
    if len(sys.argv) > 1:
        m.neighbour_streams = int(sys.argv[1])
    inp = random_inputs(B, H, W, 1002)
    d = {k: v.cuda() for k, v in inp.items() if k != "gumbel_u"}
    noise = [u.cuda() for u in inp["gumbel_u"]]
    marks = []

    def wrap(name, fn):
        def f(*a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **kw)
            e1.record()
            marks.append((name, e0, e1))
            return r
        return f

    m._feature_extraction = wrap("feature extraction", m._feature_extraction)
    m._trunk = wrap("trunk", m._trunk)
    run = lambda: m(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"], None, gumbel_uniform=noise)
    with torch.no_grad():
        run(); run()
        marks.clear()
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record(); run(); s1.record()
    torch.cuda.synchronize()
    tot = s0.elapsed_time(s1)
    (n1, a0, a1), (n2, b0, b1) = marks
    print(f"forward {tot:.1f} ms: stems {s0.elapsed_time(a0):.1f} | {n1} {a0.elapsed_time(a1):.1f} | neighbours + fusion "
          f"{a1.elapsed_time(b0):.1f} | {n2} {b0.elapsed_time(b1):.1f} | upsampler tail {b1.elapsed_time(s1):.1f}")


if __name__ == "__main__":
    main()
