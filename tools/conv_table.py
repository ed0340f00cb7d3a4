"""Developer tool (GPU box only): per-shape table of every convolution launch of one CVSR_V8 forward (c3 shape)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdfo_amd import kernels as K
from arch.SIDECVSR_our import CVSR_V8
from _inputs import random_inputs

rec = []


def wrap(name, fn, sig):
    def f(*a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **kw)
        e1.record()
        rec.append((sig(*a, **kw), e0, e1))
        return r
    return f


def conv_sig(srcs, pc, **kw):
    if isinstance(srcs, torch.Tensor):
        srcs = [srcs]
    s = srcs[0]
    return (f"conv{pc.ks}x{pc.ks} s{kw.get('stride', 1)} {pc.Cin}->{pc.Cout} {s.shape[1]}x{s.shape[2]} B{s.shape[0]} "
            f"prec{kw.get('prec', 0)} src{'16' if s.dtype == torch.float16 else '32'} "
            f"{'s2d ' if kw.get('s2d') else ''}{'o16 ' if kw.get('out_f16') else ''}{'res ' if kw.get('res1') is not None else ''}"
            f"{'shuf ' if pc.shuffle2 else ''}{'mask' if pc.tap_mask is not None else ''}", 2.0 * s.shape[0] * s.shape[1] * s.shape[2]
            * pc.Cin * pc.Cout * pc.ks * pc.ks / (kw.get('stride', 1) ** 2) * (4 / 9 if pc.tap_mask is not None else 1))


def ws_sig(src, pc, **kw):
    return (f"ws 64->{pc.Cout} {src.shape[2]}x{src.shape[3]} B{src.shape[0]} {'s2d' if kw.get('s2d') else ''}",
            2.0 * src.shape[0] * src.shape[2] * src.shape[3] * 64 * pc.Cout * 9)


def main():
    B, H, W = 8, 272, 480
    m = CVSR_V8()
    m = m.cuda().eval()
    inp = random_inputs(B, H, W, 1002)
    d = {k: v.cuda() for k, v in inp.items() if k != "gumbel_u"}
    noise = [u.cuda() for u in inp["gumbel_u"]]
    run = lambda: m(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"], None, gumbel_uniform=noise)
    with torch.no_grad():
        run()
        torch.cuda.synchronize()
        K.conv = wrap("conv", K.conv, conv_sig)
        K.conv3x3_ws = wrap("ws", K.conv3x3_ws, ws_sig)
        run()
        torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for (sig, fl), e0, e1 in rec:
        t = e0.elapsed_time(e1)
        a = agg.setdefault(sig, [0, 0.0, fl])
        a[0] += 1
        a[1] += t
    tot = sum(a[1] for a in agg.values())
    print(f"# total conv time {tot:.1f} ms over {len(rec)} launches (event-bracketed, includes launch gaps)")
    for sig, (n, t, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{t:8.2f} ms {n:4d}x {t / n:7.3f} ms/launch {fl / (t / n) / 1e9:7.1f} TF/s  {sig}")


if __name__ == "__main__":
    main()
