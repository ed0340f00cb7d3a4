"""Developer probe (GPU box only): s_memtime stamps inside one steady-state row step of the row-streaming qkv kernel, per wave."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cdfo_amd import kernels as K, _lib

B, H, W = 56, 272, 480
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.randn(B, H, W, 64, device="cuda", generator=g)
wq = torch.randn(192, 64, 1, 1, device="cuda", generator=g) / 8
wd = torch.randn(192, 1, 3, 3, device="cuda", generator=g) / 3
gamma, beta = torch.rand(64, device="cuda", generator=g) + 0.5, torch.randn(64, device="cuda", generator=g) * 0.1
packed = K.pack_qkv_dw(wq, gamma, beta)
K.qkv_dw(x, packed, wd, gram=True)
clk = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")
lib = _lib.lib()
lib.cdfo_qkv_dw_probe.argtypes = [ctypes.c_void_p]
lib.cdfo_qkv_dw_probe.restype = None
lib.cdfo_qkv_dw_probe(ctypes.c_void_p(clk.data_ptr()))
for _ in range(3):
    K.qkv_dw(x, packed, wd, gram=True)
torch.cuda.synchronize()
lib.cdfo_qkv_dw_probe(None)
c = clk.view(256, 8, 8).cpu().double()
ok = c[:, :, 7] > 0
names = ["step entry", "LayerNorm -> fragment row", "MFMAs -> y ring", "barrier passed", "y columns read", "depthwise, v staged", "Gram exchange + sums", "v row stored (end)"]
print(f"waves with stamps: {int(ok.sum())}; step length, cycles (100 MHz s_memtime ticks x clock ratio NOT applied): median {(c[:, :, 7] - c[:, :, 0])[ok].median():.0f}")
d = (c[:, :, 1:] - c[:, :, :-1])[ok]
for k in range(7):
    print(f"   {names[k]:28s} -> {names[k + 1]:28s} {d[:, k].median():8.0f}   (p10 {d[:, k].quantile(0.1):.0f}, p90 {d[:, k].quantile(0.9):.0f})")
# spread of the waves' arrival at the barrier inside a workgroup
arr = c[:, :, 2]
sp = (arr.max(1).values - arr.min(1).values)[ok.all(1)]
print(f"barrier arrival spread inside a workgroup: median {sp.median():.0f} (p90 {sp.quantile(0.9):.0f})")
