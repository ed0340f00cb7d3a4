"""Developer tool (GPU box only): the fused DCNv2 forward (cdfo_dcn_forward) at the alignment module's c3 shape
(C = Co = 64, dg = 16, 3x3, 272x480) against its HBM roofline: algorithmic bytes = (C + 3*dg*k*k + Co) * P * 4 per image
+ the weights, HIP-event timed on the launch stream.
    python tools/bench_dcn.py [--batch 8] [--iters 20]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdfo_amd.dcn import modulated_deform_conv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--height", type=int, default=272)
    ap.add_argument("--width", type=int, default=480)
    ap.add_argument("--offsets", default="mv", choices=["mv", "random"],
                    help="mv: the alignment module's kind of field (arch.py:3345-3347) -- a motion vector constant on 8x8 "
                         "blocks, |mv| <= 3 px, shared by all taps, plus a 0.5 px per-tap residual; random: independent "
                         "N(0, 3 px) per tap and pixel (worst case for the gathers)")
    ap.add_argument("--backward", action="store_true", help="time cdfo_dcn_backward (all five gradients) instead")
    a = ap.parse_args()
    B, C, Co, H, W, dg = a.batch, 64, 64, a.height, a.width, 16
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(B, C, H, W, device="cuda", generator=g)
    w = torch.randn(Co, C, 3, 3, device="cuda", generator=g) / 24
    b = torch.randn(Co, device="cuda", generator=g)
    if a.offsets == "random":
        off = 3.0 * torch.randn(B, 2 * dg * 9, H, W, device="cuda", generator=g)
    else:
        mv = (torch.rand(B, 2, (H + 7) // 8, (W + 7) // 8, device="cuda", generator=g) * 6 - 3)
        mv = mv.repeat_interleave(8, 2).repeat_interleave(8, 3)[:, :, :H, :W]
        off = mv.repeat(1, dg * 9, 1, 1) + 0.5 * torch.randn(B, 2 * dg * 9, H, W, device="cuda", generator=g)
    msk = torch.rand(B, dg * 9, H, W, device="cuda", generator=g)
    if a.backward:
        from cdfo_amd import deform_conv_cuda as ext
        go = torch.randn(B, Co, H, W, device="cuda", generator=g)
        e = torch.empty(0, device="cuda")
        gi, gw, gb = torch.zeros_like(x), torch.zeros_like(w), torch.zeros_like(b)
        goff, gm = torch.zeros_like(off), torch.zeros_like(msk)
        run = lambda: ext.modulated_deform_conv_cuda_backward(x, w, b, e, off, msk, e, gi, gw, gb, goff, gm, go, 3, 3, 1, 1, 1, 1,  # noqa: E731
                                                              1, 1, 1, dg, True)
        run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        P = H * W
        # each operand read once, each gradient written once (grad_input also read: it is accumulated into)
        bytes_ = ((C + 3 * dg * 9 + Co) + (2 * C + 3 * dg * 9)) * P * 4 * B + 2 * w.numel() * 4
        print(f"dcn_bwd offsets={a.offsets} B={B} {H}x{W} C=Co=64 dg=16: {ms:.3f} ms/launch  {bytes_ / ms / 1e6:.0f} GB/s algorithmic "
              f"({bytes_ / ms / 1e6 / 8000 * 100:.1f} % of 8 TB/s)  {3 * 2.0 * C * Co * 9 * P * B / ms / 1e9:.1f} TFLOP/s fp32")
        return
    with torch.no_grad():
        for _ in range(3):
            modulated_deform_conv(x, off, msk, w, b, 1, 1, 1, 1, dg)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            modulated_deform_conv(x, off, msk, w, b, 1, 1, 1, 1, dg)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    P = H * W
    bytes_ = (C + 3 * dg * 9 + Co) * P * 4 * B + w.numel() * 4
    flops = 2.0 * C * Co * 9 * P * B
    print(f"dcn_fwd offsets={a.offsets} B={B} {H}x{W} C=Co=64 dg=16: {ms:.3f} ms/launch  {bytes_ / ms / 1e6:.0f} GB/s algorithmic "
          f"({bytes_ / ms / 1e6 / 8000 * 100:.1f} % of 8 TB/s)  {flops / ms / 1e9:.1f} TFLOP/s fp32")


if __name__ == "__main__":
    main()
