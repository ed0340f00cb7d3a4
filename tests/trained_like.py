"""Test helper (GPU): a TRAINED-LIKE ``state_dict`` for ``CVSR_V8``, produced by this repository's own HIP training path.

The reference ships no checkpoint (README.md:27-29: external download), so every parity number of the earlier rounds was
measured on random-init weights.  This helper runs the loop of train_LD_37.py:359-381 -- Adam, the six-argument call,
Charbonnier loss, backward, step -- for `steps` iterations on synthetic super-resolution clips (a smooth random scene, seven
frames displaced by a global motion, area-downsampled x4 with sensor noise; the coding priors as in
oracle.cvsr_v8_ref.make_inputs) and returns the weights on the CPU.  It is fully described by (weight seed, data seed, steps,
learning rate): nothing is stored."""
import numpy as np
import torch
import torch.nn.functional as F


def _scene(gen, B, Hh, Wh, device):
    """Smooth random luma field in [0, 1] with edges: low-passed noise + a few hard rectangles."""
    z = torch.rand(B, 1, Hh // 8 + 2, Wh // 8 + 2, device=device, generator=gen)
    s = F.interpolate(z, size=(Hh + 32, Wh + 32), mode="bicubic", align_corners=False).clamp(0, 1)
    for b in range(B):
        for _ in range(4):
            y0, x0 = (int(v) for v in torch.randint(0, min(Hh, Wh), (2,), device=device, generator=gen))
            s[b, :, y0:y0 + 24, x0:x0 + 40] = torch.rand((), device=device, generator=gen)
    return s


def synthetic_sr_batch(gen, B, H, W, device):
    """(model inputs dict, HR target [B,1,4H,4W]): the centre frame's scene is the target; frame k sees it shifted by (k-3)*d."""
    Hh, Wh = 4 * H, 4 * W
    sc = _scene(gen, B, Hh, Wh, device)
    d = torch.randint(-3, 4, (B, 2), device=device, generator=gen)            # HR pixels per frame step
    frames = []
    for k in range(7):
        rows = []
        for b in range(B):
            dy, dx = int(d[b, 0]) * (k - 3), int(d[b, 1]) * (k - 3)
            rows.append(sc[b:b + 1, :, 16 + dy:16 + dy + Hh, 16 + dx:16 + dx + Wh])
        frames.append(torch.cat(rows, 0))
    hr = frames[3]
    lr = torch.stack([F.avg_pool2d(f, 4) for f in frames], 1)                                  # [B,7,1,H,W]
    lr = (lr + torch.randn(lr.shape, device=device, generator=gen) * (2.0 / 255.0)).clamp(0, 1)
    lr = torch.round(lr * 255.0) / 255.0
    u8 = lambda *s: torch.randint(0, 256, s, device=device, generator=gen).float() / 255.0  # noqa: E731
    rms = (torch.randn(B, 1, 7, H, W, device=device, generator=gen) * 6).round().clamp(-128, 127) / 255.0
    scale = torch.tensor([3., 2., 1., 0., -1., -2., -3.], device=device).view(1, 7, 1, 1, 1)
    mv = (d.flip(1).float() / 4.0).view(B, 1, 2, 1, 1) * scale * torch.ones(1, 1, 1, H, W, device=device)   # (x, y) in LR pixels
    return dict(x=lr, mvs0=-mv, mvs1=mv.contiguous(), pms=u8(B, 7, 1, H, W), rms=rms, ufs=u8(B, 1, 7, H, W)), hr


def trained_like_state_dict(wseed=0, data_seed=4000, steps=200, lr=2e-4, B=4, H=32, W=32):
    from arch.SIDECVSR_our import CVSR_V8
    from oracle.cvsr_v8_ref import make_state_dict
    dev = torch.device("cuda", 0)
    m = CVSR_V8(SCGs=8)
    m.load_state_dict(make_state_dict(wseed, perturb=False), strict=True)
    m = m.to(dev).train()
    opt = torch.optim.Adam(m.parameters(), lr=lr)
    gen = torch.Generator(device=dev).manual_seed(data_seed)
    torch.manual_seed(data_seed)                 # the Gumbel noise keys of the forwards
    losses = []
    for _ in range(steps):
        d, hr = synthetic_sr_batch(gen, B, H, W, dev)
        opt.zero_grad(set_to_none=True)
        sr, _ = m(d["x"], d["mvs0"], d["mvs1"], d["pms"], d["rms"], d["ufs"])
        loss = torch.sum(torch.sqrt((sr - hr) ** 2 + 1e-4)) / sr.numel()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    assert all(torch.isfinite(v).all() for v in sd.values()) and np.isfinite(losses).all()
    return sd, losses
