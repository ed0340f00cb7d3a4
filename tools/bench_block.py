"""Developer micro-benchmark: the kernels of one trunk Block_ at c3 size, fp16x2 mode (GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdfo_amd import kernels as K
from cdfo_amd.cvsr_v8 import CVSR_V8

B, H, W = 8, 272, 480
m = CVSR_V8().cuda().eval()
w = m._weights()
p = "recon_trunk.body.0.body.0."
b0, b2, dn, up, fused = w[p + "body.0"], w[p + "body.2"], w[p + "down.0"], w[p + "up.0"], w[p + "down_fused"]
x = torch.randn(B, H, W, 64, device="cuda")


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        r = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, r


def show(name, fn, flops):
    ms, r = timeit(fn)
    print(f"{name:44s} {ms:7.3f} ms  {flops/ms/1e9:7.1f} TF/s", flush=True)
    return r


P = B * H * W
t = show("1x conv1 64->256 fp16x1 -> fp16", lambda: K.conv(x, b0, pad=1, act=1, prec=K.PREC_FP16X1, out_f16=True), 2.0 * P * 9 * 64 * 256)
out = show("1x conv2 256->64 fp16 src + res", lambda: K.conv(t, b2, pad=1, res1=x), 2.0 * P * 9 * 64 * 256)
xd = show("down2(x)", lambda: K.resample2(x, up=False), 0)
d = show("half dn 1x1", lambda: K.conv(xd, dn, prec=1), 2.0 * P / 4 * 64 * 64)
td = show("half conv1", lambda: K.conv(d, b0, pad=1, act=1, prec=K.PREC_FP16X1, out_f16=True), 2.0 * P / 4 * 9 * 64 * 256)
dd = show("half conv2", lambda: K.conv(td, b2, pad=1), 2.0 * P / 4 * 9 * 64 * 256)
du = show("half up 1x1", lambda: K.conv(dd, up, prec=1), 2.0 * P / 4 * 64 * 64)
show("up2 accumulate into out", lambda: K.resample2(du, up=True, out=out, accumulate=True), 0)
u1 = show("1x up 1x1", lambda: K.conv(x, up, prec=1), 2.0 * P * 64 * 64)
u = show("up2 -> fp16 [2H,2W,64]", lambda: K.resample2(u1, up=True, out_f16=True), 0)
ts = show("2x conv1 64->256 fp16 src -> s2d fp16", lambda: K.conv(u, b0, pad=1, act=1, s2d=True, out_f16=True), 2.0 * 4 * P * 9 * 64 * 256)
show("composed sparse conv 1024->64 fp16 src + res", lambda: K.conv(ts, fused, pad=1, res1=out), 2.0 * 4 * P * 9 * 64 * 256)
g = w["recon_trunk.body.0.conv"]
show("group conv 64->64 fp16x2 + res", lambda: K.conv(x, g, pad=1, res1=x, prec=K.PREC_FP16X2), 2.0 * P * 9 * 64 * 64)
