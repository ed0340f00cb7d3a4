"""Developer benchmark (GPU box): the prior U-net's 16-channel layers (arch/SIDECVSR_our.py:1815-1834) stand-alone at c3 sizes
(56 frames of 272x480), one line per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arch.SIDECVSR_our import CVSR_V8
from cdfo_amd import kernels as K


def timed(name, f, n=10):
    for _ in range(3):
        out = f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        out = f()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:34s} {e0.elapsed_time(e1) / n:7.3f} ms -> {tuple(out.shape)}")
    return out


def main():
    torch.manual_seed(0)
    m = CVSR_V8().cuda().eval()
    w = m._weights()
    raw = w["raw"]
    u = "transformer_feature_extraction.path1.side_to_feaoneUDSA.body."
    B, H, W = 56, 272, 480
    prior = torch.rand(B, H, W, device="cuda")
    t0 = timed("udsa_head (1 -> 16, composed)", lambda: K.udsa_head(prior, H * W, B, H, W, w["udsa_head"]))
    t2 = timed("body.2 conv s2 272x480", lambda: K.small_conv16(t0, raw[u + "2.weight"], raw[u + "2.bias"], 2, 2, act=K.ACT_LRELU))
    t4 = timed("body.4 conv s2 137x241", lambda: K.small_conv16(t2, raw[u + "4.weight"], raw[u + "4.bias"], 2, 2, act=K.ACT_LRELU))
    t6 = timed("body.6 spatial gate 70x122", lambda: K.spatial_gate16(t4, raw[u + "6.spatial.weight"], raw[u + "6.spatial.bias"]))
    t7 = timed("body.7 convT s2 70x122", lambda: K.small_conv16(t6, raw[u + "7.weight"], raw[u + "7.bias"], 2, 2, 0, True, K.ACT_LRELU))
    t9 = timed("body.9 convT s2 137x241 (hi|lo)", lambda: K.small_conv16(t7, raw[u + "9.weight"], raw[u + "9.bias"], 2, 2, 1, True, K.ACT_LRELU, out_hl=True))
    res = torch.randn(B, H, W, 64, device="cuda")
    timed("body.11 16 -> 64 ring + res", lambda: K.conv_ring(t9, w[u + "11_hl"], act=K.ACT_LRELU, res1=res, plane_wrap=2))
    timed("body.0 n16 (64 -> 16)", lambda: K.conv3x3_n16(res, w[u + "0_n16"], raw[u + "0.bias"], K.ACT_LRELU))


if __name__ == "__main__":
    main()
