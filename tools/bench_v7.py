"""Developer tool (GPU box only): CVSR_V7 forward at the c3 frame size -- frames/s and the event-bracketed time of every
cdfo_amd.kernels call (grouped by function + shapes), single stream.
    python tools/bench_v7.py [--batch 2] [--steps 3] [--precision bf16x3]"""
import argparse, collections, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdfo_amd import kernels as K
from cdfo_amd import deform_conv_cuda as D
from arch.SIDECVSR_our import CVSR_V7
from _inputs import random_inputs

rec, depth = [], [0]


def shp(a):
    if isinstance(a, torch.Tensor):
        return "x".join(map(str, a.shape))
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
        rec.append((name + " " + " ".join(s for s in map(shp, a) if s) + " " + " ".join(
            k for k, v in kw.items() if v is not None and v is not False and k != "out"), e0, e1))
        return r
    return f


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--height", type=int, default=272)
    ap.add_argument("--width", type=int, default=480)
    ap.add_argument("--precision", default="bf16x3")
    ap.add_argument("--streams", type=int, default=3, help="side streams for the neighbour pipelines (1 = single stream)")
    a = ap.parse_args()
    B, H, W = a.batch, a.height, a.width
    torch.manual_seed(0)
    m = CVSR_V7()                                   # random init of the reference architecture
    m = m.cuda().eval()
    m.precision = a.precision
    m.neighbour_streams = a.streams
    inp = random_inputs(B, H, W, 1002, levels=(2, 1, 0))
    d = {k: v for k, v in inp.items() if k != "gumbel_u"}
    noise = inp["gumbel_u"]
    run = lambda: m(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"], None, gumbel_uniform=noise)
    with torch.no_grad():
        run()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        import time
        t0.record()
        h0 = time.perf_counter()
        for _ in range(a.steps):
            run()
        host_ms = (time.perf_counter() - h0) * 1e3 / a.steps
        t1.record()
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / a.steps
        print(f"# host issue time {host_ms:.1f} ms/forward")
        print(f"# CVSR_V7 {a.precision} B={B} {H}x{W}: {ms:.1f} ms/forward = {B / ms * 1e3:.2f} frames/s; "
              f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
        for mod in (K, D):
            for name in dir(mod):
                fn = getattr(mod, name)
                if isinstance(fn, types.FunctionType) and not name.startswith("_") and fn.__module__ == mod.__name__ \
                        and not name.startswith("pack") and name not in ("empty_act",):
                    setattr(mod, name, wrap(name, fn))
        run()
        torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for sig, e0, e1 in rec:
        x = agg.setdefault(sig, [0, 0.0])
        x[0] += 1
        x[1] += e0.elapsed_time(e1)
    tot = sum(x[1] for x in agg.values())
    print(f"# bracketed K calls {tot:.1f} ms over {len(rec)} calls")
    for sig, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        print(f"{t:8.2f} ms {n:4d}x {t / n:7.3f} ms/call  {sig[:150]}")


if __name__ == "__main__":
    main()
