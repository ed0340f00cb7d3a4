"""Developer benchmark (GPU box only): eager forward vs CVSR_V8.capture replay at c2 (4 x 120x240) and c3 (8 x 272x480)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arch.SIDECVSR_our import CVSR_V8
from _inputs import random_inputs


def main():
    torch.manual_seed(0)
    m = CVSR_V8().cuda().eval()
    for B, H, W, n in ((4, 120, 240, 30), (8, 272, 480, 10), (1, 272, 480, 30)):
        d = {k: v.cuda() for k, v in random_inputs(B, H, W, 3).items() if k != "gumbel_u"}
        args = (d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"])
        with torch.no_grad():
            for _ in range(3):
                m(*args)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                m(*args)
            torch.cuda.synchronize()
            te = (time.perf_counter() - t0) / n
            m.range_guard = False
            for _ in range(2):
                m(*args)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                m(*args)
            torch.cuda.synchronize()
            tn = (time.perf_counter() - t0) / n
            m.range_guard = True
            cap = m.capture(*args)
            for _ in range(3):
                cap.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                cap.replay()
            torch.cuda.synchronize()
            tg = (time.perf_counter() - t0) / n
            t0 = time.perf_counter()
            for _ in range(n):
                cap(*args)
            torch.cuda.synchronize()
            tc = (time.perf_counter() - t0) / n
        print(f"{B}x{H}x{W}: eager {1e3 * te:.2f} ms ({B / te:.1f} frames/s), eager without the range guard {1e3 * tn:.2f}, graph replay {1e3 * tg:.2f} ms "
              f"({B / tg:.1f} frames/s), replay + input copies {1e3 * tc:.2f}", flush=True)
        del cap, d, args
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
