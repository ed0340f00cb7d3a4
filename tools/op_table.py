"""Developer tool (GPU box only): event-bracketed time of every cdfo_amd.kernels call of one single-stream CVSR_V8
forward at the c3 shape (--batch N clips, --streaming = the cached-feature call), grouped by (function, tensor shapes).  Nested wrappers (a call made by another K function)
are charged to the outermost call only."""
import sys, os, collections, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdfo_amd import kernels as K
from arch.SIDECVSR_our import CVSR_V8
from _inputs import random_inputs

rec, depth = [], [0]


def shp(a):
    if isinstance(a, torch.Tensor):
        return "x".join(map(str, a.shape)) + ("h" if a.dtype == torch.float16 else "")
    if isinstance(a, (list, tuple)) and a and isinstance(a[0], torch.Tensor):
        return "[" + ",".join(shp(t) for t in a) + "]"
    if hasattr(a, "Cin") and hasattr(a, "Cout"):
        return f"W{a.Cin}->{a.Cout}k{getattr(a, 'ks', '?')}"
    return None


def wrap(name, fn):
    def f(*a, **kw):
        if depth[0]:
            return fn(*a, **kw)
        depth[0] += 1
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        try:
            r = fn(*a, **kw)
        finally:
            depth[0] -= 1
        e1.record()
        sig = name + " " + " ".join(s for s in map(shp, a) if s) + " " + " ".join(
            f"{k}" for k, v in kw.items() if v is not None and v is not False and k not in ("out",))
        rec.append((sig, e0, e1))
        return r
    return f


def main():
    B, H, W = 8, 272, 480
    streaming = "--streaming" in sys.argv
    if "--batch" in sys.argv:
        B = int(sys.argv[sys.argv.index("--batch") + 1])
    m = CVSR_V8()
    m = m.cuda().eval()
    m.neighbour_streams = 1
    inp = random_inputs(B, H, W, 1002)
    d = {k: v.cuda() for k, v in inp.items() if k != "gumbel_u"}
    noise = [u.cuda() for u in inp["gumbel_u"]]
    pre = [None]
    run = lambda: m(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"], pre[0], gumbel_uniform=noise)
    with torch.no_grad():
        _, L1 = run()
        if streaming:
            pre[0] = L1
            run()
        torch.cuda.synchronize()
        for name in dir(K):
            fn = getattr(K, name)
            if isinstance(fn, types.FunctionType) and not name.startswith("_") and fn.__module__ == K.__name__ \
                    and not name.startswith("pack") and name not in ("empty_act", "from_cp16", "sparse_taps_f16"):
                setattr(K, name, wrap(name, fn))
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        run()
        t1.record()
        torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for sig, e0, e1 in rec:
        a = agg.setdefault(sig, [0, 0.0])
        a[0] += 1
        a[1] += e0.elapsed_time(e1)
    tot = sum(a[1] for a in agg.values())
    print(f"# forward {t0.elapsed_time(t1):.1f} ms; bracketed K calls {tot:.1f} ms over {len(rec)} calls")
    for sig, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{t:8.2f} ms {n:4d}x {t / n:7.3f} ms/call  {sig}")


if __name__ == "__main__":
    main()
