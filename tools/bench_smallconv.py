"""Developer tool (GPU box only): the prior U-net's 16-channel convolutions at the bench shape (56 frames of 272x480)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdfo_amd import kernels as K


def t(fn, n=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


g = torch.Generator(device="cuda").manual_seed(1)
B = 56
w = torch.randn(16, 16, 3, 3, device="cuda", generator=g) / 12
b = torch.randn(16, device="cuda", generator=g)
x0 = torch.randn(B, 272, 480, 16, device="cuda", generator=g)
x1 = K.small_conv16(x0, w, b, 2, 2, act=K.ACT_LRELU)
x2 = K.small_conv16(x1, w, b, 2, 2, act=K.ACT_LRELU)
x3 = K.small_conv16(x2, w, b, 2, 2, 0, True, K.ACT_LRELU)
print("shapes", tuple(x1.shape), tuple(x2.shape), tuple(x3.shape))
print(f"conv s2 272x480 -> 137x241: {t(lambda: K.small_conv16(x0, w, b, 2, 2, act=K.ACT_LRELU)):.3f} ms")
print(f"conv s2 137x241 -> 70x122:  {t(lambda: K.small_conv16(x1, w, b, 2, 2, act=K.ACT_LRELU)):.3f} ms")
print(f"convT s2 70x122 -> 137x241: {t(lambda: K.small_conv16(x2, w, b, 2, 2, 0, True, K.ACT_LRELU)):.3f} ms")
print(f"convT s2 137x241 -> 272x480 (fp32): {t(lambda: K.small_conv16(x3, w, b, 2, 2, 1, True, K.ACT_LRELU)):.3f} ms")
print(f"convT s2 137x241 -> 272x480 (hi|lo planes): {t(lambda: K.small_conv16(x3, w, b, 2, 2, 1, True, K.ACT_LRELU, out_hl=True)):.3f} ms")
print(f"spatial gate 70x122: {t(lambda: K.spatial_gate16(x2, torch.randn(1, 2, 7, 7, device='cuda') / 7, torch.zeros(1, device='cuda'))):.3f} ms")
