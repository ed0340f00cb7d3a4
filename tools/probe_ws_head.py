"""Developer probe (GPU box): conv_offset_mask_ws against conv_offset_mask (single-pass tiled kernel), error by channel and phase."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdfo_amd import kernels as K

def main():
    B, H, W = 2, 24, 36
    g = torch.Generator(device="cuda").manual_seed(3)
    wt = torch.randn(432, 64, 3, 3, device="cuda", generator=g) / 24
    bs = torch.randn(432, device="cuda", generator=g) * 0.1
    pc = K.pack_conv(wt, bs)
    o1, o2 = (torch.randn(B, H, W, 64, device="cuda", generator=g) for _ in range(2))
    flow = torch.randn(B, 2, H, W, device="cuda", generator=g) * 3
    h1, h2 = K.to_cp16(o1), K.to_cp16(o2)
    r1, r2 = (h.permute(0, 2, 3, 1, 4).reshape(B, H, W, 64).float().contiguous() for h in (h1, h2))
    print("to_cp16 round trip", (r1 - o1).abs().max().item())
    for phase in (0, 1):
        off_t, mask_t = torch.zeros(B, 288, H, W, device="cuda"), torch.zeros(B, 144, H, W, device="cuda")
        off_w, mask_w = torch.zeros_like(off_t), torch.zeros_like(mask_t)
        if phase == 1:
            for t in (off_t, off_w):
                t.copy_(torch.randn(t.shape, device="cuda", generator=g))
            m0 = torch.randn(mask_t.shape, device="cuda", generator=g)
            mask_t.copy_(m0); mask_w.copy_(m0)
            off_w.copy_(off_t)
        K.conv_offset_mask(r1, pc, off_t, mask_t, flow, 10.0, bool(phase), K.PREC_FP16X1)
        K.conv_offset_mask_ws(h1, pc, off_w, mask_w, flow, 10.0, bool(phase))
        torch.cuda.synchronize()
        eo = (off_w - off_t).abs().amax(dim=(0, 2, 3))
        em = (mask_w - mask_t).abs().amax(dim=(0, 2, 3))
        print("phase", phase, "offset max err", eo.max().item(), "mask max err", em.max().item())
        bad = (eo > 1e-3).nonzero().flatten().tolist()
        print(" bad offset channels:", bad[:40], "count", len(bad))
        badm = (em > 1e-3).nonzero().flatten().tolist()
        print(" bad mask channels:", badm[:40], "count", len(badm))
        if bad:
            c = bad[0]
            e = (off_w - off_t)[0, c].abs()
            print(" channel", c, "bad pixel rows", (e.amax(1) > 1e-3).nonzero().flatten().tolist()[:30], "cols", (e.amax(0) > 1e-3).nonzero().flatten().tolist()[:40])
            print(" sample ws", off_w[0, c, 0, :6].tolist(), "tiled", off_t[0, c, 0, :6].tolist())

if __name__ == "__main__":
    main()
