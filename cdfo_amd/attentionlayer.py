"""``DSTA`` -- the deformable spatial attention gate of the reference (ops/attentionlayer.py:86-156; the class is defined
twice there, byte-identically), with the same constructor, parameter names and ``forward(x) -> x * m * y``.
Every operator runs in libcdfo_hip.so: the thin 16-channel convs / pooling / resizing through the NCHW helpers of
csrc/nchw_ops.hip, the DCNv2 through the fused ``cdfo_dcn_forward`` (no CPU fallback).  With gradients enabled the same
operator graph runs as ``torch.autograd.Function``s (``cdfo_amd/nchw_autograd.py``: the gather kernels of csrc/nchw_bwd.hip,
``cdfo_dcn_backward`` for the deformable convolution), so the module trains like the reference's."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import nchw as N
from . import nchw_autograd as G
from .dcn import ModulatedDeformConv
from .kernels import ACT_NONE, ACT_RELU, ACT_SIGMOID, on_device


class DSTA(nn.Module):
    def __init__(self, n_feats):
        super().__init__()
        f = n_feats // 4
        self.conv1 = nn.Conv2d(n_feats, f, kernel_size=1)
        self.conv_f = nn.Conv2d(f, f, kernel_size=1)
        self.conv_max = nn.Conv2d(f, f, kernel_size=3, padding=1)
        self.conv2 = nn.Conv2d(f, f, kernel_size=3, stride=2, padding=0)
        self.conv3 = nn.Conv2d(f, f, kernel_size=3, padding=1)
        self.conv3_ = nn.Conv2d(f, f, kernel_size=3, padding=1)
        self.conv4 = nn.Conv2d(f, n_feats, kernel_size=1)
        self.sigmoid = nn.Sigmoid()
        self.relu = nn.ReLU(inplace=True)
        self.dcn = ModulatedDeformConv(f, f, 3, padding=1, deformable_groups=f)
        self.mask = nn.Conv2d(f, f * 3 * 3 * 3, 3, padding=1)
        self.f = f
        self.down_conv2 = nn.Sequential(nn.Conv2d(f, f, 3, stride=2, padding=3 // 2), nn.ReLU(inplace=True))
        self.mask2 = nn.Conv2d(f, f * 3 * 3 * 3, 3, padding=1)
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.conv_du = nn.Sequential(nn.Conv2d(f, 2 * f, 1, padding=0, bias=True), nn.ReLU(inplace=True),
                                     nn.Conv2d(2 * f, n_feats, 1, padding=0, bias=True), nn.Sigmoid())

    @staticmethod
    def _c(m: nn.Conv2d, x, act=ACT_NONE):
        return N.conv2d(x, m.weight.detach(), None if m.bias is None else m.bias.detach(), m.stride[0], m.padding[0], act)

    def forward(self, x):
        if not x.is_cuda:
            raise NotImplementedError("DSTA (HIP): CPU tensors are not supported")
        with on_device(x):
            if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
                return self._forward_train(x)
            return self._forward(x)

    def _forward_train(self, x):
        """ops/attentionlayer.py:117-156 operator by operator under autograd (HIP kernels in both directions)."""
        cv = lambda m, t, act=ACT_NONE: G.conv2d(t, m.weight, m.bias, m.stride[0], m.padding[0], act)  # noqa: E731
        c1_ = cv(self.conv1, x)
        c1 = cv(self.conv2, c1_)
        v_max = G.maxpool(c1, 7, 3)
        v_range = cv(self.conv_max, v_max, ACT_RELU)
        c3 = cv(self.conv3, v_range, ACT_RELU)
        c3 = cv(self.conv3_, c3, ACT_RELU)
        dc3 = cv(self.down_conv2[0], c3, ACT_RELU)
        off_mask2 = cv(self.mask2, dc3)
        off_msk = cv(self.mask, c3)
        off_msk = G.add(off_msk, G.resize_bilinear(off_mask2, off_msk.size(2), off_msk.size(3)))
        k2 = self.f * 2 * 3 * 3
        off, msk = off_msk[:, :k2], G.sigmoid(off_msk[:, k2:])
        c3 = G.relu(self.dcn(v_max, off, msk))
        y = G.plane_mean(c3)                                        # [B, f, 1, 1]: the gate MLP is parameter-sized torch arithmetic
        y = torch.sigmoid(torch.nn.functional.conv2d(torch.relu(torch.nn.functional.conv2d(y, self.conv_du[0].weight, self.conv_du[0].bias)),
                                                     self.conv_du[2].weight, self.conv_du[2].bias))
        c3 = G.resize_bilinear(c3, x.size(2), x.size(3))
        c4 = cv(self.conv4, G.add(c3, cv(self.conv_f, c1_)))
        return G.gate(c4, x, y)

    def _forward(self, x):
        c1_ = self._c(self.conv1, x)
        c1 = self._c(self.conv2, c1_)
        v_max = N.maxpool(c1, 7, 3)
        v_range = self._c(self.conv_max, v_max, ACT_RELU)
        c3 = self._c(self.conv3, v_range, ACT_RELU)
        c3 = self._c(self.conv3_, c3, ACT_RELU)
        dc3 = self._c(self.down_conv2[0], c3, ACT_RELU)
        off_mask2 = self._c(self.mask2, dc3)
        off_msk = self._c(self.mask, c3)
        N.resize_bilinear(off_mask2, off_msk.size(2), off_msk.size(3), out=off_msk, accumulate=True)
        k2 = self.f * 2 * 3 * 3
        off = off_msk[:, :k2].contiguous()
        msk = N.ew(off_msk[:, k2:].contiguous(), 2)
        c3 = N.ew(self.dcn(v_max, off, msk), 1)
        y = self._c(self.conv_du[2], self._c(self.conv_du[0], N.avgpool(c3), ACT_RELU), ACT_SIGMOID)
        c3 = N.resize_bilinear(c3, x.size(2), x.size(3))
        cf = self._c(self.conv_f, c1_)
        c4 = self._c(self.conv4, N.ew(c3, 0, b=cf))
        return N.ew(c4, 3, x=x, y=y)
