"""GPU parity of the LLongRangAttention kernels (seq_attn in its MFMA and VALU forms, window form) against a plain
torch-cpu softmax(QQ^T)V on the same seeded inputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(q, v, mode):
    B, H, W, C = q.shape
    if mode == 0:
        a = (q @ q.transpose(-1, -2)).softmax(-1)                      # [B,H,W,W]
        return a @ v
    if mode == 1:
        qt, vt = q.transpose(1, 2), v.transpose(1, 2)                  # [B,W,H,C]
        a = (qt @ qt.transpose(-1, -2)).softmax(-1)
        return (a @ vt).transpose(1, 2)
    qw = q.view(B, H // 8, 8, W // 8, 8, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H // 8, W // 8, 64, C)
    vw = v.view(B, H // 8, 8, W // 8, 8, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H // 8, W // 8, 64, C)
    o = (qw @ qw.transpose(-1, -2)).softmax(-1) @ vw
    return o.view(B, H // 8, W // 8, 8, 8, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, C)


@pytest.mark.parametrize("H,W", [(8, 8), (16, 40), (40, 136), (136, 72)])
@pytest.mark.parametrize("mode", [0, 1, 2, 10, 11, 12])
def test_seq_attn(H, W, mode):
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(H * 100 + W + mode)
    q = torch.randn(2, H, W, 64, generator=g) * 0.5
    v = torch.randn(2, H, W, 64, generator=g)
    ref = _ref(q, v, mode % 10)
    out = K.seq_attn(q.cuda(), v.cuda(), mode)
    torch.cuda.synchronize()
    err = (out.cpu() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("C,H,W", [(192, 16, 24), (192, 9, 13), (64, 8, 8)])
def test_dwconv3x3(C, H, W):
    import torch.nn.functional as F
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(C + H + W)
    x = torch.randn(2, C, H, W, generator=g)
    w = torch.randn(C, 1, 3, 3, generator=g)
    ref = F.conv2d(x, w, padding=1, groups=C)
    out = K.dwconv3x3(x.permute(0, 2, 3, 1).contiguous().cuda(), w.cuda())
    torch.cuda.synchronize()
    assert (out.permute(0, 3, 1, 2).cpu() - ref).abs().max().item() < 1e-5


@pytest.mark.parametrize("B,H,W", [(2, 12, 30), (1, 17, 45), (3, 8, 64), (1, 5, 7), (2, 6, 121)])
def test_qkv_dw_fused(B, H, W):
    """LayerNorm -> 1x1 (64->192) -> depthwise 3x3 in one kernel vs the same chain in torch-cpu fp32."""
    import torch.nn.functional as F
    from cdfo_amd import kernels as K
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + W)
    x = torch.randn(B, 64, H, W, generator=g) * 2 + 0.3
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    wq = torch.randn(192, 64, 1, 1, generator=g) / 8.0
    wd = torch.randn(192, 1, 3, 3, generator=g) / 3.0
    ln = F.layer_norm(x.permute(0, 2, 3, 1), (64,), gamma, beta, 1e-5).permute(0, 3, 1, 2)
    ref = F.conv2d(F.conv2d(ln, wq), wd, padding=1, groups=192)
    packed = K.pack_qkv_dw(wq.cuda(), gamma.cuda(), beta.cuda())
    out = K.qkv_dw(x.permute(0, 2, 3, 1).contiguous().cuda(), packed, wd.cuda().contiguous())
    torch.cuda.synchronize()
    err = (out.cpu().permute(0, 3, 1, 2) - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err
    # fused Gram mode: v only + the per-head sums sum_p q k^T, sum q^2, sum k^2 in cdfo_gram_partial's layout
    v, part, n = K.qkv_dw(x.permute(0, 2, 3, 1).contiguous().cuda(), packed, wd.cuda().contiguous(), gram=True)
    v2, part2, _ = K.qkv_dw(x.permute(0, 2, 3, 1).contiguous().cuda(), packed, wd.cuda().contiguous(), gram=True)
    assert torch.equal(part, part2) and torch.equal(v, v2)          # no atomics: bit-reproducible
    torch.cuda.synchronize()
    assert (v.cpu().permute(0, 3, 1, 2) - ref[:, 128:]).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    q, k = ref[:, :64].double().reshape(B, 8, 8, H * W), ref[:, 64:128].double().reshape(B, 8, 8, H * W)
    want = torch.zeros(B, 64, 10, dtype=torch.float64)
    want[:, :, :8] = (q @ k.transpose(-1, -2)).reshape(B, 64, 8)
    want[:, :, 8] = (q * q).sum(-1).reshape(B, 64)
    want[:, :, 9] = (k * k).sum(-1).reshape(B, 64)
    got = part.cpu().double().sum(1).view(B, 64, 10)
    assert (got - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item()), (got - want).abs().max().item()
