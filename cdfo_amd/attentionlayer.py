"""``DSTA`` -- the deformable spatial attention gate of the reference (ops/attentionlayer.py:86-156; the class is defined
twice there, byte-identically), with the same constructor, parameter names and ``forward(x) -> x * m * y``.
Every operator runs in libcdfo_hip.so: the thin 16-channel convs / pooling / resizing through the NCHW helpers of
csrc/nchw_ops.hip, the DCNv2 through the fused ``cdfo_dcn_forward`` (no CPU fallback; forward only)."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import nchw as N
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
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("DSTA (HIP): forward only -- wrap the call in torch.no_grad()")
        with on_device(x):
            return self._forward(x)

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
