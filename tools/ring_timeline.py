"""Developer tool (GPU box only): s_memtime stamps inside the LDS-DMA ring kernel (dbg 16, conv3x3_ring.hip) -- where a chunk's
cycles go, per wave.  Prints, for one workgroup, each wave's stamps of chunks 8..11 relative to the workgroup's earliest stamp of
chunk 8, and the mean length of every phase over all workgroups (100 MHz real-time ticks -> ns).
    python tools/ring_timeline.py [sparse|dense]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdfo_amd import kernels as K

NAMES = ["entry", "waited", "barrier", "pre-tap", "tapA", "tapB", "tapC", "tapD", "end"]


def main():
    sparse = (sys.argv[1] if len(sys.argv) > 1 else "sparse") == "sparse"
    Cin, Cout, H, W, B = (1024, 64, 272, 480, 8) if sparse else (256, 64, 272, 480, 8)
    x = torch.randn(B, H, W, Cin, device="cuda").half()
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / (Cin * (4 if sparse else 9)) ** 0.5
    masks = []
    if sparse:
        for c in range(Cin // 16):
            y0, x0 = (c >> 5) & 1, (c >> 4) & 1
            m = 0
            keep = torch.zeros(3, 3, device="cuda")
            for dy in range(2):
                for dx in range(2):
                    m |= 1 << ((y0 + dy) * 3 + x0 + dx)
            keep[y0:y0 + 2, x0:x0 + 2] = 1
            w[:, c * 16:(c + 1) * 16] *= keep
            masks.append(m)
    pc = K.pack_conv(w, torch.randn(Cout, device="cuda"))
    if sparse:
        pc.tap_mask = torch.tensor(masks, dtype=torch.int32, device="cuda")
    src = x.view(B, H, W, Cin // 16, 16).permute(0, 3, 1, 2, 4).contiguous()
    res = torch.randn(B, H, W, Cout, device="cuda")
    nwg = 256
    clk = torch.zeros(nwg, 8, 4, 10, dtype=torch.int64, device="cuda")
    out = K.conv_ring(src, pc, res1=res)
    for _ in range(3):
        K.conv_ring(src, pc, res1=res, out=out, res2=clk, dbg=16)
    torch.cuda.synchronize()
    raw = clk.cpu().double()
    nck = Cin // 16
    span = raw[:, :4, 1, 9] - raw[:, :4, 0, 9]           # consumers: entry of the next tile's last chunk - entry of its chunk 2
    if (span > 0).any():
        print(f"# undisturbed chunk period (next tile: entry of chunk 2 -> entry of chunk {nck - 1}, consumer waves): "
              f"{(span[span > 0] / (nck - 1 - 2)).mean():.1f} ticks")
    ep = raw[:, :4, 3, 9] - raw[:, :4, 2, 9]
    if (ep > 0).any():
        print(f"# epilogue of that tile (consumer waves): {ep[ep > 0].mean():.0f} ticks; last-chunk entry -> epilogue start "
              f"{(raw[:, :4, 2, 9] - raw[:, :4, 1, 9])[ep > 0].mean():.0f} ticks")
    t = raw[..., :9]                                     # [wg, wave, chunk, stamp]
    ok = (t[..., 0] > 0).all(dim=-1).all(dim=-1)
    print(f"# {'four-tap' if sparse else 'dense'} form, {int(ok.sum())} workgroups with stamps; s_memtime ticks (constant 100 MHz clock): 1 tick = 10 ns")
    t = t[ok]
    d = t[..., 1:] - t[..., :-1]                         # phase lengths
    print("# mean phase length over workgroups x waves x chunks (ticks):")
    for i in range(8):
        print(f"   {NAMES[i]:8s} -> {NAMES[i + 1]:8s} {d[..., i].mean():7.2f}   (waves 0-3 {d[:, :4, :, i].mean():6.2f}, waves 4-7 {d[:, 4:, :, i].mean():6.2f})")
    per_chunk = (t[:, :, 1:, 0] - t[:, :, :-1, 0]).mean()
    print(f"# chunk period (entry to entry): {per_chunk:.2f} ticks")
    g = t[0]
    base = g[:, 0, 0].min()
    print("# workgroup 0: stamps relative to its first chunk-8 entry (ticks)")
    for wv in range(8):
        for c in range(4):
            print(f"   wave {wv} chunk {8 + c}: " + " ".join(f"{int(v - base):5d}" for v in g[wv, c]))


if __name__ == "__main__":
    main()
