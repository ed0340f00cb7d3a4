"""Developer micro-benchmark (GPU box only): LDS-DMA ring kernel vs the tiled 16-bit kernel on Block_'s wide-input convs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdfo_amd import kernels as K
from bench_conv import timeit

# Cin, Cout, H, W, B, sparse
SHAPES = [(256, 64, 272, 480, 8, False), (256, 64, 136, 240, 8, False), (1024, 64, 272, 480, 8, True)]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dbgs = [int(d) for d in sys.argv[2].split(",")] if len(sys.argv) > 2 and sys.argv[2] else []
    only = sys.argv[3] if len(sys.argv) > 3 else ""          # "sparse" / "dense": that form only (counter passes)
    for (Cin, Cout, H, W, B, sparse) in SHAPES:
        if (only == "sparse" and not sparse) or (only == "dense" and (sparse or H != 272)):
            continue
        x = torch.randn(B, H, W, Cin, device="cuda").half()
        w = torch.randn(Cout, Cin, 3, 3, device="cuda") / (Cin * (4 if sparse else 9)) ** 0.5
        masks = None
        if sparse:
            masks = []
            for c in range(Cin // 16):
                y0, x0 = (c >> 5) & 1, (c >> 4) & 1
                m = 0
                keep = torch.zeros(3, 3, device="cuda")
                for dy in range(2):
                    for dx in range(2):
                        m |= 1 << ((y0 + dy) * 3 + x0 + dx)
                keep[y0:y0 + 2, x0:x0 + 2] = 1
                w[:, c * 16:(c + 1) * 16] *= keep
                masks.append(m)
        b = torch.randn(Cout, device="cuda")
        res = torch.randn(B, H, W, Cout, device="cuda")
        pc = K.pack_conv(w, b)
        if sparse:
            pc.tap_mask = torch.tensor(masks, dtype=torch.int32, device="cuda")
        src = x.view(B, H, W, Cin // 16, 16).permute(0, 3, 1, 2, 4).contiguous()      # chunk-planar copy of x
        fl = 2.0 * B * H * W * Cin * Cout * (4 if sparse else 9)
        ref = K.conv([x], pc, pad=1, res1=res)
        ms0 = timeit(lambda: K.conv([x], pc, pad=1, res1=res, out=ref))
        out = K.conv_ring(src, pc, res1=res)
        ms1 = timeit(lambda: K.conv_ring(src, pc, res1=res, out=out))
        worst = 0.0
        for _ in range(reps):
            out.zero_()
            K.conv_ring(src, pc, res1=res, out=out)
            worst = max(worst, (out - ref).abs().max().item())
        line = (f"{Cin}->{Cout} {H}x{W} B{B} sparse={int(sparse)}: tiled {ms0:6.3f} ms {fl/ms0/1e9:6.1f} TF/s | ring {ms1:6.3f} ms "
                f"{fl/ms1/1e9:6.1f} TF/s | max diff over {reps} runs {worst:.2e} |")
        for d in dbgs:
            ms = timeit(lambda: K.conv_ring(src, pc, res1=res, out=out, dbg=d))
            line += f" dbg{d} {ms:6.3f}"
        if sparse:       # the four-tap form with its whole epilogue: half-resolution residual (bilinear x2) + fp16 chunk-planar copy
            up = torch.randn(B, H // 2, W // 2, Cout, device="cuda")
            o16 = torch.empty(B, Cout // 16, H, W, 16, device="cuda", dtype=torch.float16)
            ms = timeit(lambda: K.conv_ring(src, pc, res1=res, out=out, res_up2=up, out2_cp16=o16))
            line += f" | full epilogue {ms:6.3f} ms {fl/ms/1e9:6.1f} TF/s"
        else:
            o16 = torch.empty(B, Cout // 16, H, W, 16, device="cuda", dtype=torch.float16)
            ms = timeit(lambda: K.conv_ring(src, pc, res1=res, out=out, out2_cp16=o16))
            line += f" | + cp16 copy {ms:6.3f} ms"
        print(line, flush=True)


if __name__ == "__main__":
    main()
