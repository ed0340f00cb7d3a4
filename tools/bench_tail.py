"""Developer benchmark (GPU box): the up-sampler tail (upconv2 + conv_last's tap sums, then the tap gather + x4 skip) stand-alone at c3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arch.SIDECVSR_our import CVSR_V8
from cdfo_amd import kernels as K

def main():
    torch.manual_seed(0)
    m = CVSR_V8().cuda().eval()
    w = m._weights()
    raw = w["raw"]
    B, H, W = 8, 272, 480
    t1 = torch.randn(B, 2 * H, 2 * W, 64, device="cuda")
    xc = torch.rand(B, 7, 1, H, W, device="cuda")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    f = lambda: K.upconv_last(t1, w["upconv2"], raw["conv_last.weight"], raw["conv_last.bias"], xc[:, 3], 7 * H * W)
    for _ in range(3):
        out = f()
    ev[0].record()
    for _ in range(10):
        out = f()
    ev[1].record()
    torch.cuda.synchronize()
    print(f"upconv_last + taps gather, {B}x{2*H}x{2*W}x64 -> {tuple(out.shape)}: {ev[0].elapsed_time(ev[1]) / 10:.3f} ms per call")

if __name__ == "__main__":
    main()
