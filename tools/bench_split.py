"""Developer experiment (GPU box): the trunk on the whole batch (every persistent kernel fills the chip) against the trunk of two
half-batches on two streams, each limited to a share of the CUs (kernels.cu_limit): does a matrix-bound kernel of one half run
beside an HBM-bound kernel of the other?   python tools/bench_split.py [share ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arch.SIDECVSR_our import CVSR_V8
from cdfo_amd import kernels as K


def timed(f, n=3):
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    shares = [int(a) for a in sys.argv[1:]] or [128, 160, 192]
    torch.manual_seed(0)
    m = CVSR_V8().cuda().eval()
    m.H, m.W = 272, 480
    w = m._weights()
    x = torch.randn(8, 272, 480, 64, device="cuda") * 0.5
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    main_s = torch.cuda.current_stream()
    with torch.no_grad():
        ref = m._trunk(w, x)
        print(f"whole batch, all CUs (side stream for the half-resolution branch on): {timed(lambda: m._trunk(w, x)):.2f} ms")
        m.trunk_side_stream = False
        print(f"whole batch, all CUs, one stream: {timed(lambda: m._trunk(w, x)):.2f} ms")
        xa, xb = x[:4].contiguous(), x[4:].contiguous()
        print(f"half batch alone, all CUs: {timed(lambda: m._trunk(w, xa)):.2f} ms")
        for share in shares:
            def split():
                s1.wait_stream(main_s); s2.wait_stream(main_s)
                with K.cu_limit(share):
                    with torch.cuda.stream(s1):
                        ya = m._trunk(w, xa)
                    with torch.cuda.stream(s2):
                        yb = m._trunk(w, xb)
                main_s.wait_stream(s1); main_s.wait_stream(s2)
                return ya, yb
            ya, yb = split()
            torch.cuda.synchronize()
            err = max((ya - ref[:4]).abs().max().item(), (yb - ref[4:]).abs().max().item())
            print(f"two half batches on two streams, {share} CUs each: {timed(split):.2f} ms (max |diff| vs whole batch {err:.1e})")
            with K.cu_limit(share):
                print(f"   half batch alone on {share} CUs: {timed(lambda: m._trunk(w, xa)):.2f} ms")


if __name__ == "__main__":
    main()
