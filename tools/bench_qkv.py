"""Developer tool (GPU box only): the fused LayerNorm + qkv 1x1 + depthwise 3x3 + Gram kernel at the bench shape (56 frames)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdfo_amd import kernels as K


def main():
    B, H, W = 56, 272, 480
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(B, H, W, 64, device="cuda", generator=g)
    wq = torch.randn(192, 64, 1, 1, device="cuda", generator=g) / 8
    wd = torch.randn(192, 1, 3, 3, device="cuda", generator=g) / 3
    gamma, beta = torch.rand(64, device="cuda", generator=g) + 0.5, torch.randn(64, device="cuda", generator=g) * 0.1
    packed = K.pack_qkv_dw(wq, gamma, beta)
    for gram in (True, False):
        run = lambda: K.qkv_dw(x, packed, wd, gram=gram)
        for _ in range(2):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record()
        torch.cuda.synchronize()
        print(f"qkv_dw gram={gram}: {e0.elapsed_time(e1) / 5:.3f} ms")


if __name__ == "__main__":
    main()
