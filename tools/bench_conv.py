"""Developer micro-benchmark: time the conv kernels on the trunk's shapes with HIP events (GPU box only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdfo_amd import kernels as K

SHAPES = [  # Cin, Cout, H, W, B
    (64, 256, 544, 960, 8), (256, 64, 544, 960, 8), (64, 256, 272, 480, 8), (256, 64, 272, 480, 8),
    (64, 64, 272, 480, 8), (64, 256, 136, 240, 8), (256, 64, 136, 240, 8)]


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    precs = [int(p) for p in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 2]
    for (Cin, Cout, H, W, B) in SHAPES:
        x = torch.randn(B, H, W, Cin, device="cuda")
        w = torch.randn(Cout, Cin, 3, 3, device="cuda") / (Cin * 9) ** 0.5
        b = torch.randn(Cout, device="cuda")
        pc = K.pack_conv(w, b)
        out = K.empty_act(B, H, W, Cout, x.device)
        fl = 2.0 * B * H * W * Cin * Cout * 9
        line = f"{Cin:3d}->{Cout:3d} {H}x{W} B{B}:"
        for prec in precs:
            ms = timeit(lambda: K.conv([x], pc, pad=1, act=1, out=out, prec=prec))
            line += f"  prec{prec}: {ms:7.3f} ms {fl/ms/1e9:7.1f} TF/s"
        print(line, flush=True)


if __name__ == "__main__":
    main()
