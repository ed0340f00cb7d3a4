"""Developer micro-benchmark: 1x1 conv shapes of the CVSR_V8 forward (GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdfo_amd import kernels as K

P = 272 * 480
SHAPES = [  # name, pixels, [source channel counts], Cout, residual?, shuffle?
    ("qkv 64->192 (56 frames)", 56 * P, [64], 192, False, False),
    ("apply 64->64 +res (56 frames)", 56 * P, [64], 64, True, False),
    ("trunk 64->64 @P", 8 * P, [64], 64, False, False),
    ("trunk 64->64 @P/4", 2 * P, [64], 64, False, False),
    ("input_conv 64->128", 8 * P, [64], 128, False, False),
    ("fuse 128->64 +res", 8 * P, [128], 64, True, False),
    ("fusion 64+64->64", 8 * P, [64, 64], 64, False, False),
    ("folded 3x64->64", 8 * P, [64, 64, 64], 64, False, False),
    ("tsa 7x64->64", 8 * P, [64] * 7, 64, False, False),
    ("upconv1 64->256 shuffle", 8 * P, [64], 256, False, True),
    ("upconv2 64->256 shuffle @4P", 32 * P, [64], 256, False, True),
]


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, npix, srcs, Cout, res, shuf in SHAPES:
    W = 480
    H = npix // W
    xs = [torch.randn(1, H, W, c, device="cuda") for c in srcs]
    cin = sum(srcs)
    w = torch.randn(Cout, cin, 1, 1, device="cuda") / cin ** 0.5
    pc = K.pack_conv(w, torch.randn(Cout, device="cuda"), shuffle2=shuf)
    r = torch.randn(1, H, W, Cout, device="cuda") if res else None
    out = None if shuf else K.empty_act(1, H, W, Cout, "cuda")
    by = 4.0 * npix * (cin + Cout * (2 if res else 1))
    line = f"{name:34s}"
    for prec in (0, 1):
        ms = timeit(lambda: K.conv(xs, pc, act=1, res1=r, out=out, prec=prec))
        line += f"  prec{prec}: {ms:7.3f} ms {by/ms/1e6:7.1f} GB/s"
    print(line, flush=True)
