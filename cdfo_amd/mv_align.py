"""``MVDualAttAlignment`` -- the motion-vector-guided deformable alignment of the reference's ``CVSR_V7``
(arch/SIDECVSR_our.py:3265-3352, instantiated at :4242 as ``MVDualAttAlignment(64, 64, 3, padding=1,
deformable_groups=16, max_residue_magnitude=10)``): same constructor, parameter names and
``forward(x, extra_feat, pred_feat, flow_1)``; NCHW tensors in and out like the reference.  Everything runs in
libcdfo_hip.so: MV warp, the shared 8-head channel attention folded into a per-image 64x64 matrix, the 64->64->432
offset/mask head on the MFMA conv kernels, offset assembly (10*tanh + flipped MV) and the fused DCNv2
(``torchvision.ops.deform_conv2d`` in the reference, arch.py:3352).  No CPU fallback.  With gradients enabled the forward runs
operator by operator as ``torch.autograd.Function``s (``forward_train``: the pixel-major Functions of ``cdfo_amd/autograd.py``,
the layout / offset-assembly Functions of ``cdfo_amd/nchw_autograd.py`` and the DCN operator's own Function), so the module
trains like the reference's; the folded inference schedule is used under ``torch.no_grad()``."""
from __future__ import annotations

import ctypes as C
import os

import torch
import torch.nn as nn

from . import _lib, deform_conv_cuda
from . import kernels as K
from ._lib import check
from .dcn import ModulatedDeformConvPack


class MVDualAttAlignment(ModulatedDeformConvPack):
    def __init__(self, *args, **kwargs):
        self.max_residue_magnitude = kwargs.pop('max_residue_magnitude', 10)
        super().__init__(*args, **kwargs)
        self.conv_offset = nn.Sequential(
            nn.Conv2d(self.out_channels, self.out_channels, 3, 1, 1),
            nn.LeakyReLU(negative_slope=0.1, inplace=True),
            nn.Conv2d(self.out_channels, 27 * self.deformable_groups, 3, 1, 1),
        )
        dim = 64
        self.num_heads = 8
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.conv_du = nn.Sequential(
            nn.Conv2d(self.out_channels, self.out_channels // 16, 1, padding=0, bias=True), nn.ReLU(inplace=True),
            nn.Conv2d(self.out_channels // 16, self.out_channels, 1, padding=0, bias=True), nn.Sigmoid())
        self.fusion_out = nn.Conv2d(dim * 2, dim, kernel_size=1, bias=False)
        self.temperature = nn.Parameter(torch.ones(self.num_heads, 1, 1))
        self.project_out = nn.Conv2d(dim, dim, kernel_size=1, bias=False)
        self.sigmoid = nn.Sigmoid()
        nn.init.constant_(self.conv_offset[-1].weight, 0)
        nn.init.constant_(self.conv_offset[-1].bias, 0)
        # arithmetic of the 64->64->432 offset/mask head (the module's FLOPs): "bf16x3" = split-bf16 matrix cores,
        # fp32-grade (~1e-6 relative; the head's output is scaled by 10 px, so nothing coarser), or "f32" = exact
        self.precision = "bf16x3"
        self.fuse_assembly = os.environ.get("CDFO_V7_FUSE_ASSEMBLY", "1") != "0"      # developer A/B switches
        self.head_one_pass = os.environ.get("CDFO_V7_HEAD_1PASS", "1") != "0"
        self.off0_one_pass = os.environ.get("CDFO_V7_OFF0_1PASS", "1") != "0"
        self.ws_head = os.environ.get("CDFO_V7_WS_HEAD", "1") != "0"
        self._packed = None
        self._sig = None

    def _weights(self):
        sig = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._packed is None or sig != self._sig:
            d = lambda t: t.detach().contiguous()  # noqa: E731
            self._packed = dict(
                fusion=K.pack_conv(d(self.fusion_out.weight), None),
                off0=K.pack_conv(d(self.conv_offset[0].weight), d(self.conv_offset[0].bias)),
                off2=K.pack_conv(d(self.conv_offset[2].weight), d(self.conv_offset[2].bias)))
            self._sig = sig
        return self._packed

    def forward(self, x, extra_feat, pred_feat, flow_1):
        if not x.is_cuda:
            raise NotImplementedError("MVDualAttAlignment (HIP): CPU tensors are not supported")
        if self.in_channels != 64 or self.out_channels != 64:
            raise NotImplementedError("MVDualAttAlignment (HIP): specialised for 64 channels (arch.py:4242)")
        if torch.is_grad_enabled() and (any(t.requires_grad for t in (x, extra_feat, pred_feat))
                                        or any(p.requires_grad for p in self.parameters())):
            with K.on_device(x):
                return self.forward_train(x, extra_feat, pred_feat, flow_1)
        x = x.contiguous().float()
        with K.on_device(x):
            return self.forward_pm(x, K.nchw_to_nhwc(x), K.nchw_to_nhwc(extra_feat.float()),
                                   K.nchw_to_nhwc(pred_feat.float()), flow_1.contiguous().float())

    def forward_train(self, x, extra_feat, pred_feat, flow_1):
        """arch.py:3303-3352 under autograd: NCHW tensors in and out; gradients reach x, extra_feat, pred_feat and every parameter
        the reference's forward uses (the motion field is a network input).  Convolution arithmetic = autograd.CONV_PREC."""
        import torch.nn.functional as F
        from . import autograd as A
        from . import nchw_autograd as G
        from .dcn import modulated_deform_conv
        x = x.float()
        flow = flow_1.detach().contiguous().float()
        B, _, H, W = x.shape
        xq, extra, pred = G.to_pixel_major(x), G.to_pixel_major(extra_feat.float()), G.to_pixel_major(pred_feat.float())
        warped = A.flow_warp(extra, flow, 2 * H * W)
        fused = A.conv([warped, pred], self.fusion_out.weight)                     # no activation here (arch.py:3305)
        w0, b0, w2, b2 = (self.conv_du[0].weight, self.conv_du[0].bias, self.conv_du[2].weight, self.conv_du[2].bias)
        heads = []
        for v in (warped, pred):
            gate = torch.sigmoid(F.linear(F.relu(F.linear(A.chan_mean(v), w0.flatten(1), b0)), w2.flatten(1), b2))   # [B, 64]
            att = A.channel_attention(xq, fused, A.scale_channels(v, gate), self.temperature, self.num_heads)
            o = A.conv(att, self.project_out.weight)
            o = A.conv(o, self.conv_offset[0].weight, self.conv_offset[0].bias, 1, 1, K.ACT_LRELU)
            heads.append(A.conv(o, self.conv_offset[2].weight, self.conv_offset[2].bias, 1, 1))      # [B, H, W, 27 dg]
        offset, mask = G.offset_mask(heads[0], heads[1], flow, 9 * self.deformable_groups, self.max_residue_magnitude)
        return modulated_deform_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation,
                                     self.groups, self.deformable_groups)

    def forward_pm(self, x, xq, extra, pred, flow):
        """The same computation for callers that already hold pixel-major tensors (``CVSR_V7``): ``x`` NCHW (the DCN's
        input), ``xq`` = the same features pixel-major [B,H,W,64], ``extra`` / ``pred`` pixel-major, ``flow`` NCHW
        [B,2,H,W] contiguous.  Returns the aligned features NCHW."""
        w = self._weights()
        B, _, H, W = x.shape
        P = H * W
        d = lambda t: t.detach().contiguous()  # noqa: E731
        prec = {"bf16x3": K.PREC_BF16X3, "fp16x2": K.PREC_FP16X2}.get(self.precision, K.PREC_F32)
        warped = K.flow_warp(extra, flow, 2 * P)
        p1 = K.PREC_F32 if prec == K.PREC_F32 else K.PREC_BF16X3        # the 1x1 convolutions: split-bf16 (fp32-grade) on the matrix cores
        fused = K.conv([warped, pred], w["fusion"], prec=p1)             # no activation here (arch.py:3305)
        gp, ng = K.gram_partial(xq, fused, 8)
        fold = K.mdta_fold(gp, ng, d(self.temperature), d(self.project_out.weight))   # P . blockdiag(softmax)
        third = 9 * self.deformable_groups
        offset = torch.empty((B, 2 * third, H, W), dtype=torch.float32, device=x.device)
        mask = torch.empty((B, third, H, W), dtype=torch.float32, device=x.device)
        # the 16-bit head writes the DCN's NCHW offset / mask planes from its own epilogue (first head: tanh + flow / raw mask sums,
        # second head: in-place accumulate + sigmoid); the exact-fp32 head and widths that are no multiple of 4 assemble separately
        fuse_epilogue = self.fuse_assembly and prec != K.PREC_F32 and W % 4 == 0
        # the weights-stationary head addresses its source and the offset planes of ONE image with 32-bit buffer offsets
        # (cdfo_conv3x3_c64_ws_offmask): larger frames (e.g. 1920x1080 at dg = 16) take the tiled head below
        ws_head = (self.ws_head and prec == K.PREC_FP16X2 and self.head_one_pass and self.off0_one_pass and H % 2 == 0
                   and (2 * third) % 32 == 0 and K.conv_offset_mask_ws_fits(B, H, W, third))
        fuse_epilogue = fuse_epilogue or ws_head
        outs = []
        for n_head, v in enumerate((warped, pred)):
            part, n = K.chan_sum_partial(v)
            gate = K.vec_mlp(part, n, P, d(self.conv_du[0].weight), d(self.conv_du[0].bias), self.out_channels // 16,
                             K.ACT_RELU, d(self.conv_du[2].weight), d(self.conv_du[2].bias), 64, K.ACT_SIGMOID)
            if ws_head:      # the 1x1 kernel also writes the fp16 chunk-planar copy the weights-stationary kernel reads
                _, o = K.conv(v, K.fold_scale_inputs(fold, gate), prec=p1, cp16_out=True)
            else:
                o = K.conv(v, K.fold_scale_inputs(fold, gate), prec=p1)   # project_out(attn @ (v * gate)), the gate folded into the matrix
            if ws_head:
                # both convolutions of the head on the weights-stationary kernel of CVSR_V8's trunk (single-pass fp16, the same
                # rounding points as the tiled kernel's single-pass mode): ~2x its rate at 64 -> 432
                o = K.conv3x3_ws(o, w["off0"], act=K.ACT_LRELU)
                K.conv_offset_mask_ws(o, w["off2"], offset, mask, flow, self.max_residue_magnitude, n_head == 1)
                continue
            o = K.conv(o, w["off0"], pad=1, act=K.ACT_LRELU,
                       prec=K.PREC_FP16X1 if (prec == K.PREC_FP16X2 and self.off0_one_pass) else prec)
            if fuse_epilogue:
                # fp16x2: the head's weights are rounded once to fp16 in that mode; rounding its input once as well (one MFMA pass
                # instead of two) adds an error of the same size -- CVSR_V7's parity moves inside its 10x margin
                hp = K.PREC_FP16X1 if (prec == K.PREC_FP16X2 and self.head_one_pass) else prec
                K.conv_offset_mask(o, w["off2"], offset, mask, flow, self.max_residue_magnitude, n_head == 1, hp)
            else:
                outs.append(K.conv(o, w["off2"], pad=1, prec=prec))       # [B,H,W,27*dg]
        if not fuse_epilogue:
            vp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
            check(_lib.lib().cdfo_mv_offset_mask(vp(outs[0]), vp(outs[1]), outs[0].stride(2), vp(flow), C.c_longlong(2 * P),
                                                 B, C.c_longlong(P), third, float(self.max_residue_magnitude), vp(offset),
                                                 vp(mask), K._stream()), "cdfo_mv_offset_mask")
        out = torch.empty((B, self.out_channels, H, W), dtype=torch.float32, device=x.device)
        deform_conv_cuda.modulated_deform_conv_cuda_forward(
            x, d(self.weight), None if self.bias is None else d(self.bias), None, offset, mask, out, None,
            self.kernel_size[0], self.kernel_size[1], self.stride, self.stride, self.padding, self.padding,
            self.dilation, self.dilation, self.groups, self.deformable_groups, self.bias is not None)
        return out
