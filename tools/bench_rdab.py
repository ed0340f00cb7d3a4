"""Developer benchmark (GPU box): the Gumbel hard-mask + 9-tap channel convolution kernel (cdfo_rdab_prep_rng, arch.py:2168-2219)
stand-alone at one neighbour of the bench shape (8 x 272 x 480); prints the time per launch and a fingerprint of its three outputs
(A/B builds through CDFO_LIB_PATH must agree except where a pixel's softmax sits within rounding of the 0.5 threshold)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdfo_amd import kernels as K


def main():
    B, H, W = 8, 272, 480
    g = torch.Generator(device="cuda").manual_seed(5)
    xq = torch.randn(B, H, W, 128, device="cuda", generator=g)
    vmax = torch.rand(B, 64, device="cuda", generator=g) * 2
    wW = torch.randn(1, 1, 1, 9, device="cuda", generator=g) * 0.3
    bW = torch.randn(1, device="cuda", generator=g) * 0.1
    noise = torch.empty(B, 64, H, W, device="cuda")
    outs = K.rdab_prep_rng(xq, vmax, 12345, 2, wW, bW, noise_out=noise)
    ref = K.rdab_prep(xq, vmax, noise, wW, bW)                      # the injected-noise entry on the captured noise: same kernel body
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(outs, ref))
    masked = (outs[2] == 0).float().mean().item()                   # share of (pixel, channel) with mask = 1 (window query zeroed)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        K.rdab_prep_rng(xq, vmax, 12345, 2, wW, bW, outs=outs)
    e0.record()
    for _ in range(20):
        K.rdab_prep_rng(xq, vmax, 12345, 2, wW, bW, outs=outs)
    e1.record()
    torch.cuda.synchronize()
    fp = [float(t.double().sum().item()) for t in outs]
    print(f"rdab_prep_rng {B}x{H}x{W}: {e0.elapsed_time(e1) / 20:.4f} ms per launch; rng == injected: {same}; masked share {masked:.5f}; "
          f"fingerprints {fp[0]:.6f} {fp[1]:.6f} {fp[2]:.6f}")


if __name__ == "__main__":
    main()
