"""Developer tool (GPU box only): the three attention products of LLongRangAttention (row / column / 8x8 window) stand-alone
at the bench shape (24 frames of 272x480, 64 channels): time per launch, MFMA rate, and the maximum difference from the
VALU reference form of the same kernel (modes 10 / 11 / 12)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdfo_amd import kernels as K


def main():
    B, H, W = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (24, 272, 480)
    g = torch.Generator(device="cuda").manual_seed(3)
    q = torch.randn(B, H, W, 64, device="cuda", generator=g) * 0.7
    v = torch.randn(B, H, W, 64, device="cuda", generator=g)
    for mode in (0, 1, 2, 20, 21, 22):
        L = (W, H, 64)[mode % 10]
        out = K.seq_attn(q, v, mode)
        ref = K.seq_attn(q[:2], v[:2], 10 + mode % 10)
        err = (out[:2] - ref).abs().max().item()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            K.seq_attn(q, v, mode, out=out)
        e0.record()
        n = 10
        for _ in range(n):
            K.seq_attn(q, v, mode, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        flop = 3 * 2 * 2 * 64.0 * B * H * W * L      # two products, three fp16 passes each
        print(f"mode {mode} L={L}: {ms:.3f} ms  {flop / ms / 1e9:.0f} TFLOP/s issued (fp16 MFMA, 3 passes)  max|mfma - valu| = {err:.2e}")


if __name__ == "__main__":
    main()
