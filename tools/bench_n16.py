"""Developer benchmark (GPU box): the prior U-net's first layer (64 -> 16, 3x3) on 56 frames of 272x480: the streaming kernel
cdfo_conv3x3_c64_n16 against the tiled 16-bit kernel it replaces on the default path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdfo_amd import kernels as K


def main():
    B, H, W = 56, 272, 480
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, H, W, 64, device="cuda", generator=g)
    w = torch.randn(16, 64, 3, 3, device="cuda", generator=g) / 24
    b = torch.randn(16, device="cuda", generator=g)
    pn, pc = K.pack_conv_n16(w), K.pack_conv(w, b)
    fns = {"n16 (16x16x32 stream)": lambda: K.conv3x3_n16(x, pn, b, K.ACT_LRELU),
           "tiled mma16 bf16x3": lambda: K.conv([x], pc, pad=1, act=K.ACT_LRELU, prec=K.PREC_BF16X3)}
    outs = {}
    for name, f in fns.items():
        for _ in range(3):
            outs[name] = f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        gb = B * H * W * (64 + 16) * 4 / 1e9
        print(f"{name}: {ms:.3f} ms per launch, {gb / ms:.2f} TB/s of algorithmic bytes")
    a, c = outs.values()
    print("max |n16 - tiled| =", (a - c).abs().max().item())


if __name__ == "__main__":
    main()
