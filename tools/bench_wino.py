"""Developer micro-benchmark (GPU box only): the Winograd F(2,3) Block_.body[0] kernel vs the direct weights-stationary one."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdfo_amd import kernels as K
from bench_conv import timeit

SHAPES = [(256, 544, 960, 8, True), (256, 272, 480, 8, False), (256, 136, 240, 8, False), (256, 272, 480, 1, False), (256, 1088, 1920, 1, True)]


DBGS = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 4, 8, 16, 6, 14, 30, 31]


def main():
    for (Cout, H, W, B, s2d) in SHAPES:
        x = torch.randn(B, H, W, 64, device="cuda")
        w = torch.randn(Cout, 64, 3, 3, device="cuda") / 24.0
        b = torch.randn(Cout, device="cuda")
        pc = K.pack_conv(w, b)
        src = K.to_cp16(x)
        fl = 2.0 * B * H * W * 64 * Cout * 9
        out = K.conv3x3_ws(src, pc, act=1, s2d=s2d)
        ms0 = timeit(lambda: K.conv3x3_ws(src, pc, act=1, s2d=s2d, out=out))
        o2 = K.conv3x3_wino(src, pc, act=1, s2d=s2d)
        ms1 = timeit(lambda: K.conv3x3_wino(src, pc, act=1, s2d=s2d, out=o2))
        d = (o2.float() - out.float()).abs().max().item()
        gb = (B * H * W * (64 + Cout) * 2) / 1e9
        line = (f"64->{Cout} {H}x{W} B{B} s2d={int(s2d)}: direct {ms0:6.3f} ms {fl/ms0/1e9:6.1f} TF/s | wino {ms1:6.3f} ms "
                f"{fl/ms1/1e9:6.1f} TF/s (algorithmic), {gb/ms1:5.2f} TB/s | max diff {d:.2e} | ablations")
        for dbg in DBGS:
            ms = timeit(lambda: K.conv3x3_wino(src, pc, act=1, s2d=s2d, out=o2, dbg=dbg))
            line += f" [{dbg}] {ms:.3f}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
