"""Developer micro-benchmark (GPU box only): weights-stationary Block_.body[0] kernel vs the tiled 16-bit kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdfo_amd import kernels as K
from bench_conv import timeit

SHAPES = [(256, 544, 960, 8, True), (256, 272, 480, 8, False), (256, 136, 240, 8, False)]


def main():
    dbgs = [int(p) for p in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0]
    for (Cout, H, W, B, s2d) in SHAPES:
        x = torch.randn(B, H, W, 64, device="cuda")
        w = torch.randn(Cout, 64, 3, 3, device="cuda") / 24.0
        b = torch.randn(Cout, device="cuda")
        pc = K.pack_conv(w, b)
        x16 = x.half()
        src = K.to_cp16(x)
        fl = 2.0 * B * H * W * 64 * Cout * 9
        ref = K.conv([x16], pc, pad=1, act=1, s2d=s2d, out_f16=True)
        ms0 = timeit(lambda: K.conv([x16], pc, pad=1, act=1, s2d=s2d, out_f16=True, out=ref))
        line = f"64->{Cout} {H}x{W} B{B} s2d={int(s2d)}: tiled {ms0:6.3f} ms {fl/ms0/1e9:6.1f} TF/s |"
        for d in dbgs:
            out = K.conv3x3_ws(src, pc, act=1, s2d=s2d, dbg=d)
            ms = timeit(lambda: K.conv3x3_ws(src, pc, act=1, s2d=s2d, out=out, dbg=d))
            line += f" ws[dbg {d}] {ms:6.3f} ms {fl/ms/1e9:6.1f} TF/s"
            if d == 0:
                line += f" (max diff vs tiled {(K.from_cp16(out).float() - ref.float()).abs().max().item():.2e})"
        clk = torch.zeros(256 * 8 * 3, dtype=torch.int64, device="cuda")
        for d in (128,):
            K.conv3x3_ws(src, pc, act=1, s2d=s2d, out=out, dbg=d, clk=clk)
            torch.cuda.synchronize()
            c = clk.view(256, 8, 3).cpu().double()
            t0 = c[:, :, 1].min()
            dur = (c[:, :, 2] - c[:, :, 1]) / 100.0                  # us per wave
            end = (c[:, :, 2] - t0) / 100.0
            start = (c[:, :, 1] - t0) / 100.0
            mhz = (c[:, :, 0] / dur).mean().item()
            wg_end = end.max(dim=1).values
            line += (f" | dbg{d}: clock {mhz:.0f} MHz; wave busy us min/mean/max {dur.min():.0f}/{dur.mean():.0f}/{dur.max():.0f};"
                     f" start max {start.max():.0f}; WG end min/mean/max {wg_end.min():.0f}/{wg_end.mean():.0f}/{wg_end.max():.0f};"
                     f" slowest XCD means {[round(wg_end[x::8].mean().item()) for x in range(8)]}")
        ms = timeit(lambda: K.to_cp16(x))
        print(line + f" | to_cp16 {ms:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
