"""Developer benchmark (GPU box only): one training step of CVSR_V8 at the reference script's batch (train_LD_37.py: 20 crops of
64x64, Charbonnier loss) -- forward and backward through the HIP autograd path, timed separately."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arch.SIDECVSR_our import CVSR_V8
from _inputs import random_inputs


def main():
    B, H, W = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (20, 64, 64)
    m = CVSR_V8().cuda().train()
    inp = random_inputs(B, H, W, 7)
    d = {k: v.cuda() for k, v in inp.items() if k != "gumbel_u"}
    hr = torch.rand(B, 1, 4 * H, 4 * W, device="cuda")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for it in range(4):
        m.zero_grad(set_to_none=True)
        ev[0].record()
        out, _ = m(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"])
        loss = torch.sum(torch.sqrt((out - hr) ** 2 + 1e-4))
        ev[1].record()
        loss.backward()
        ev[2].record()
        torch.cuda.synchronize()
        if it:
            print(f"training step {B}x{H}x{W}: forward {ev[0].elapsed_time(ev[1]):.1f} ms, backward {ev[1].elapsed_time(ev[2]):.1f} ms", flush=True)


if __name__ == "__main__":
    main()
