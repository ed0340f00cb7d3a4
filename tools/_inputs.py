"""Synthetic weights / clips for the developer tools (shapes and statistics of SURVEY section 8d), generated with torch on
the device.  The tools do not touch oracle/ (that is test infrastructure): a default-constructed model already carries a
random init of the reference architecture."""
import torch


def random_inputs(B, H, W, seed=0, device="cuda", levels=None):
    """x, pms, ufs ~ U{0..255}/255; rms ~ round(N(0, 6^2))/255; mvs0 / mvs1 = a block-constant field scaled per slot;
    gumbel_u: six U(0,1) tensors [B,64,H,W], or 36 per-level ones when ``levels`` (CVSR_V7) is given."""
    g = torch.Generator(device=device).manual_seed(seed)
    u8 = lambda *s: torch.randint(0, 256, s, device=device, generator=g).float() / 255.0  # noqa: E731
    x, pms, ufs = u8(B, 7, 1, H, W), u8(B, 7, 1, H, W), u8(B, 1, 7, H, W)
    rms = (torch.randn(B, 1, 7, H, W, device=device, generator=g) * 6).round().clamp(-128, 127) / 255.0
    scale = torch.tensor([3., 2., 1., 0., -1., -2., -3.], device=device).view(1, 7, 1, 1, 1)

    def field():
        m = torch.randint(-64, 64, (B, 2, (H + 7) // 8, (W + 7) // 8), device=device, generator=g).float() / 128.0
        return (m.repeat_interleave(8, 2).repeat_interleave(8, 3)[:, :, :H, :W].unsqueeze(1) * scale).contiguous()

    d = dict(x=x, pms=pms, ufs=ufs, rms=rms, mvs0=-field(), mvs1=field())
    noise = lambda h, w: torch.rand(B, 64, h, w, device=device, generator=g).clamp_min_(1e-6)  # noqa: E731
    if levels:
        d["gumbel_u"] = [noise(H >> lv, W >> lv) for lv in levels for _ in range(12)]
    else:
        d["gumbel_u"] = [noise(H, W) for _ in range(6)]
    return d
