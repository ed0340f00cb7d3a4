"""Developer probe: cdfo_dcn_backward's data kernel with and without the grad_input scatter (alignment shape, B = 8)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cdfo_amd import _lib
lib = _lib.lib()
B, Cc, Co, H, W, dg = 8, 64, 64, 272, 480, 16
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, Cc, H, W, device="cuda", generator=g); w = torch.randn(Co, Cc, 3, 3, device="cuda", generator=g) / 24
mv = (torch.rand(B, 2, 34, 60, device="cuda", generator=g) * 6 - 3).repeat_interleave(8, 2).repeat_interleave(8, 3)[:, :, :H, :W]
off = mv.repeat(1, dg * 9, 1, 1) + 0.5 * torch.randn(B, 2 * dg * 9, H, W, device="cuda", generator=g)
msk = torch.rand(B, dg * 9, H, W, device="cuda", generator=g); go = torch.randn(B, Co, H, W, device="cuda", generator=g)
gi, goff, gm = torch.zeros_like(x), torch.zeros_like(off), torch.zeros_like(msk)
p = lambda t: C.c_void_p(None if t is None else t.data_ptr())
def run(gin, gof, gms, name):
    f = lambda: _lib.check(lib.cdfo_dcn_backward(p(x), p(off), p(msk), p(w), p(go), p(gin), p(gof), p(gms), None, None, B, Cc, H, W, Co,
                                                 3, 3, 1, 1, 1, 1, 1, 1, 1, dg, C.c_float(1.0), None), "bwd")
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [f() for _ in range(3)]; e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 3:.3f} ms")
run(gi, goff, gm, "all three")
run(None, goff, gm, "grad_offset + grad_mask only (no scatter)")
run(gi, None, None, "grad_input only")
