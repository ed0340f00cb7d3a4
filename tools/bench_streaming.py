"""Developer benchmark (GPU box only): the reference's real evaluation scenario -- ONE sequence, one new frame per
forward, feature cache (test_LD_22_FPS.py:155-192) -- through cdfo_amd.streaming.StreamingSR on synthetic data."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from arch.SIDECVSR_our import CVSR_V8
from cdfo_amd.streaming import StreamingSR


def main():
    T, H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 24, 270, 480
    rs = np.random.RandomState(0)
    u8 = lambda: rs.randint(0, 256, size=(T, H, W)).astype(np.uint8)
    lr, pms, ufs = u8(), u8(), u8()
    rms = np.clip(np.round(rs.randn(T, H, W) * 6), -128, 127).astype(np.float32)
    mv = rs.randint(-64, 64, size=(2, T, (H + 7) // 8, (W + 7) // 8, 3)).astype(np.float32)
    mv[..., 2] = rs.choice([-2.0, -1.0, 1.0], size=mv.shape[:-1])
    mv = np.repeat(np.repeat(mv, 8, axis=2), 8, axis=3)[:, :, :H, :W]
    model = CVSR_V8()
    model = model.cuda().eval()
    # (HIP graph, neighbour streams, frames per group, group of frames 0-2 beside the new frame's feature extraction, new frame alone)
    for use_graph, nstr, grp, ovl, alone in ((False, 2, 3, False, False), (True, 2, 3, False, False), (False, 2, 3, True, False),
                                             (True, 2, 3, True, False), (False, 2, 3, True, True), (True, 2, 3, True, True),
                                             (True, 1, 3, False, False)):
        model.neighbour_streams, model.neighbour_group, model.overlap_new_frame, model.new_frame_alone = nstr, grp, ovl, alone
        s = StreamingSR(model, lr, pms, rms, ufs, mv[0], mv[1], use_graph=use_graph)
        s.run()                               # warm-up (weight packing, first-touch allocations, graph capture)
        outs = s.run()
        print(f"streaming, 1 sequence of {T} frames {H}x{W} -> {tuple(outs[0].shape)}, HIP graph {use_graph}, neighbour streams {nstr}, frames per group {grp}, overlap {ovl}, new frame alone {alone}: "
              f"{s.fps:.2f} frames/s ({1e3 * s.seconds / T:.1f} ms per frame, forward only, B=1)", flush=True)


def pipelined():
    T, H, W = 48, 270, 480
    if "--size" in sys.argv:
        k = sys.argv.index("--size")
        T, H, W = int(sys.argv[k + 1]), int(sys.argv[k + 2]), int(sys.argv[k + 3])
    rs = np.random.RandomState(0)
    u8 = lambda: rs.randint(0, 256, size=(T, H, W)).astype(np.uint8)
    lr, pms, ufs = u8(), u8(), u8()
    rms = np.clip(np.round(rs.randn(T, H, W) * 6), -128, 127).astype(np.float32)
    mv = rs.randint(-64, 64, size=(2, T, (H + 7) // 8, (W + 7) // 8, 3)).astype(np.float32)
    mv[..., 2] = rs.choice([-2.0, -1.0, 1.0], size=mv.shape[:-1])
    mv = np.repeat(np.repeat(mv, 8, axis=2), 8, axis=3)[:, :, :H, :W]
    model = CVSR_V8().cuda().eval()
    s = StreamingSR(model, lr, pms, rms, ufs, mv[0], mv[1])
    s.run()
    s.run()
    print(f"streaming, sequential loop (per-frame timing), {T} frames {H}x{W}: {s.fps:.2f} frames/s", flush=True)
    for _ in range(3):
        s.run_pipelined()
        print(f"streaming, PIPELINED over two streams (whole-loop wall time), {T} frames: {s.fps:.2f} frames/s "
              f"({1e3 * s.seconds / T:.2f} ms per frame)", flush=True)


if __name__ == "__main__":
    if "--pipelined" in sys.argv:
        pipelined()
        sys.exit(0)
    main()
