"""``CVSR_V7`` -- the reference's DCN-aligned, three-level-pyramid variant (arch/SIDECVSR_our.py:4215-4367; SURVEY
section 8f n3) with the forward pass running in libcdfo_hip.so: same class name, constructor, ``forward(x, mvs0, mvs1,
pms, rms, ufs, pre_L1_fea=None) -> (out, L1_fea)`` and the same 247 ``state_dict`` entries (names + shapes), so a
checkpoint of the reference class loads with ``load_state_dict(strict=True)``.

Structure (reference lines in the method docstrings): stems + ``PartitionTransformerBlock`` feature extraction at full
resolution, a 2x2-mean feature pyramid, per level a backward (``mvs0``) and a forward (``mvs1``) pass over the six
neighbours -- resized priors, ``RDAB`` compensation, ``conv_expand_fea_r``, ``MVDualAttAlignment`` (the fused DCNv2
kernel) -- merged by ``fb_fusion`` / ``tsa_fusion``, the cross-scale ``SCNet`` trunk on the three-level list and the
pyramid-merging upsampler.

Same deliberate differences as ``CVSR_V8``: CUDA (ROCm) tensors only (no CPU fallback); under ``torch.no_grad()`` the fused
inference schedule of this file, with gradients enabled the operator graph under autograd (``cdfo_amd/cvsr_v7_train.py``: HIP
kernels forward and backward, like the reference class the module is trainable); injectable Gumbel noise (``gumbel_uniform=`` : the 36 uniform draws of ``RDAB.gumbel_softmax`` in call order,
each ``[B,64,H>>lv,W>>lv]``), ``L1_fea`` returned channels-last, no ``featuremap_visual`` side effects.  H and W must be
multiples of 4 (two pyramid halvings; the reference's shapes only line up under the same condition).  Arithmetic:
``precision`` = "bf16x3" (split-bf16 matrix cores, fp32-grade; default), "f32" (exact), or "fp16x2" (fp16 weights,
fp16 hi+lo activations, two MFMA passes, in the 3x3 convolutions of the alignment head and ``conv_expand_fea_r``; the
trunk's body convolutions on CVSR_V8's single-pass fp16 ``Block_`` kernels; the feature extractor stays split-bf16)."""
from __future__ import annotations

import os

import contextlib
import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from . import kernels as K
from .cvsr_v8 import CVSR_V8, NF, NFRAMES, _register
from .mv_align import MVDualAttAlignment

N_DRAWS = 36


def _param_spec():
    """(key, shape, fan_in, init) of ``CVSR_V7().state_dict()`` except the ``MV_deform_align`` sub-module (arch.py:4223-4250)."""
    sp = []

    def conv(key, co, ci, k, bias=True, init="default"):
        sp.append((key + ".weight", (co, ci, k, k), ci * k * k, init))
        if bias:
            sp.append((key + ".bias", (co,), ci * k * k, "zero" if init == "kaiming0.1" else "bias"))

    conv("conv_first", 64, 1, 3)
    conv("conv_second", 64, 1, 3)
    p = "transformer_feature_extraction.path1."
    sp.append((p + "norm1.body.weight", (64,), None, "ones"))
    sp.append((p + "norm1.body.bias", (64,), None, "zero"))
    sp.append((p + "attn.temperature", (8, 1, 1), None, "ones"))
    conv(p + "attn.qkv", 192, 64, 1, bias=False)
    conv(p + "attn.qkv_dwconv", 192, 1, 3, bias=False)
    conv(p + "attn.project_out", 64, 64, 1, bias=False)
    sp.append((p + "norm2.body.weight", (64,), None, "ones"))
    sp.append((p + "norm2.body.bias", (64,), None, "zero"))
    conv(p + "conv", 64, 64, 3)
    conv(p + "SA.spatial", 1, 2, 7)
    conv("conv_expand_fea_r", 64, 128, 3)
    conv("conv_expand_ufs", 64, 1, 3)
    conv("conv_expand_rms", 64, 1, 3)
    conv("fb_fusion", 64, 128, 1)
    conv("tsa_fusion", 64, 448, 1)
    for g in range(7):
        gp = f"recon_trunk.body.{g}."
        conv(gp + "conv", 64, 64, 3)
        for b in range(3):
            bp = gp + f"body.{b}."
            conv(bp + "body.0", 256, 64, 3, init="kaiming0.1")
            conv(bp + "body.2", 64, 256, 3, init="kaiming0.1")
            conv(bp + "down.0", 64, 64, 1, init="kaiming0.1")
            conv(bp + "up.0", 64, 64, 1, init="kaiming0.1")
    conv("upconv1", 256, 84, 1)
    conv("upconv2", 256, 64, 1)
    conv("conv_last", 1, 64, 1)
    sp.append(("MV_deform_align", None, None, "module"))        # keeps the reference's registration order
    r = "RDAB."
    conv(r + "conv_du_re.0", 64, 64, 1)
    conv(r + "conv_du_re.2", 64, 64, 3)
    conv(r + "conv_du_re2.0", 64, 64, 1)
    conv(r + "conv_dc.0", 64, 64, 1)
    conv(r + "conv_dc.2", 64, 64, 1)
    conv(r + "spatial", 1, 2, 3)
    conv(r + "conv_df.0", 64, 64, 1)
    conv("upconv1_L2", 64, 64, 1)
    conv("upconv1_L3", 64, 64, 1)
    return sp


class CVSR_V7(nn.Module):
    PRECISIONS = {"f32": K.PREC_F32, "bf16x3": K.PREC_BF16X3, "fp16x2": K.PREC_FP16X2}

    def __init__(self, nf=64, nframes=7, fea_ext_RBs=7, SCGs=4, istraining=False):
        super().__init__()
        if nf != 64 or nframes != 7:
            raise ValueError("the HIP path is specialised for nf=64, nframes=7 (the only configuration the reference runs)")
        self.nf, self.center, self.istraining, self.stride = nf, nframes // 2, istraining, 4
        self.gumbel_uniform: Optional[Sequence[torch.Tensor]] = None
        self.precision = "bf16x3"
        self.neighbour_streams = 3       # HIP side streams for the twelve independent neighbour pipelines of a pyramid level
        for key, shape, fan_in, init in _param_spec():
            if init == "module":
                self.MV_deform_align = MVDualAttAlignment(64, 64, 3, padding=1, deformable_groups=16,
                                                          max_residue_magnitude=10)
                continue
            t = torch.empty(shape)
            if init == "default":
                nn.init.kaiming_uniform_(t, a=math.sqrt(5))
            elif init == "kaiming0.1":
                nn.init.kaiming_normal_(t, a=0, mode="fan_in")
                t.mul_(0.1)
            elif init == "bias":
                bound = 1.0 / math.sqrt(fan_in)
                nn.init.uniform_(t, -bound, bound)
            elif init == "ones":
                t.fill_(1.0)
            else:
                t.zero_()
            _register(self, key, nn.Parameter(t))
        self.debug_taps: Optional[dict] = None
        self._packed: Optional[dict] = None
        self._packed_sig = None

    # -- packed weights ------------------------------------------------------------------------------------------
    def _weights(self) -> dict:
        sig = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._packed is not None and sig == self._packed_sig:
            return self._packed
        sd = {k: v.detach() for k, v in self.named_parameters()}
        for v in sd.values():
            if not v.is_cuda or v.dtype != torch.float32:
                raise NotImplementedError("CVSR_V7 (HIP): fp32 parameters on the GPU expected; call .cuda()")
        w: Dict[str, object] = {}

        def pc(key, **kw):
            w[key] = K.pack_conv(sd[key + ".weight"], sd.get(key + ".bias"), **kw)

        fe = "transformer_feature_extraction.path1."
        for key in (fe + "conv", "conv_expand_fea_r", "fb_fusion", "tsa_fusion", "RDAB.conv_du_re.0", "RDAB.conv_du_re.2",
                    "RDAB.conv_dc.0", "RDAB.conv_dc.2", "RDAB.conv_df.0"):
            pc(key)
        for g in range(7):
            pc(f"recon_trunk.body.{g}.conv")
            for b in range(3):
                for leaf in ("body.0", "body.2", "down.0", "up.0"):
                    pc(f"recon_trunk.body.{g}.body.{b}.{leaf}")
        w[fe + "qkv_dw"] = K.pack_qkv_dw(sd[fe + "attn.qkv.weight"], sd[fe + "norm1.body.weight"], sd[fe + "norm1.body.bias"])
        w[fe + "conv_hl"] = K.pack_conv_hilo(sd[fe + "conv.weight"], sd[fe + "conv.bias"], True)     # split-fp16 form for conv_ring
        # RDAB's residual-mask branch as in CVSR_V8: conv_du_re.0 composed with conv_expand_rms into a second stencil of the stem
        # kernel (written in space-to-depth form), conv_du_re.2 (stride 2) as a tap-masked stride-1 convolution over that form
        w["rms_du0"] = K.compose_stem_1x1(sd["conv_expand_rms.weight"], sd["conv_expand_rms.bias"], sd["RDAB.conv_du_re.0.weight"],
                                          sd["RDAB.conv_du_re.0.bias"])
        w["RDAB.conv_du_re.2_s2d"] = K.pack_conv_s2p2_s2d(sd["RDAB.conv_du_re.2.weight"], sd["RDAB.conv_du_re.2.bias"])
        pc("upconv1_L2", shuffle2=True)
        pc("upconv2", shuffle2=True)
        dev = sd["conv_first.weight"].device
        # pixel_shuffle(pixel_shuffle(.)) of the level-3 branch (arch.py:4354): the first shuffle rides on upconv1_L3's
        # store; the second one is a 0/1 selection matrix with the shuffle store, padded from 4 to 16 output channels so
        # that the 84-channel concatenation [64 | 16 | 4] becomes three 16-aligned sources [64 | 16 | 4 + 12 zeros]
        sel = torch.zeros(64, 16, 1, 1, device=dev)
        for o in range(16):
            sel[o, o, 0, 0] = 1.0                     # out channel c*4+phase (c < 4) <- in channel c*4+phase
        w["shuffle_L3"] = K.pack_conv(sel, None, shuffle2=True)
        wu = sd["upconv1.weight"]
        wpad = torch.zeros(256, 96, 1, 1, device=dev)
        wpad[:, :84] = wu
        w["upconv1"] = K.pack_conv(wpad, sd["upconv1.bias"], shuffle2=True)
        w["upconv1_L3s"] = K.pack_conv(sd["upconv1_L3.weight"], sd["upconv1_L3.bias"], shuffle2=True)
        w3 = torch.zeros(1, 64, 3, 3, device=dev)
        w3[:, :, 1, 1] = sd["conv_last.weight"][:, :, 0, 0]           # the 1x1 conv_last as the centre tap of a 3x3
        w["conv_last3"] = w3.contiguous()
        w["raw"] = {k: v.contiguous() for k, v in sd.items()}
        self._packed, self._packed_sig = w, sig
        return w

    def _conv(self, *args, exact=False, **kw):
        """``exact``: convolutions whose result is returned to the caller (the L1_fea feature cache) stay split-bf16."""
        prec = self.PRECISIONS[self.precision]
        if exact and prec == K.PREC_FP16X2:
            prec = K.PREC_BF16X3
        return K.conv(*args, prec=prec, **kw)

    # -- building blocks ------------------------------------------------------------------------------------------
    def _feature_extraction(self, w, x1, x2):
        """``PartitionTransformerBlock.forward`` (arch.py:1350-1368): four weight-shared rounds."""
        raw = w["raw"]
        p = "transformer_feature_extraction.path1."
        lazy_gate = self.precision != "f32"          # every mode whose 1x1 convolutions run on the streaming kernel
        if lazy_gate:
            # the prior branch is only ever gated (x2_k = x2_{k-1} * g_k, one shared SpatialAttention): x2_k = x2_0 * G_k with a per-pixel
            # running product G, pool(x2_k) = G_k * pool(x2_0) -- one pooling pass, a one-float-per-pixel plane per round, and the
            # gated tensor enters the 1x1 convolution's residual sum as x2_0 * G (K.gate_map_cumulative; csrc/v7_ops.hip)
            pooled, G = K.chan_pool(x2), None
        for _ in range(4):
            if lazy_gate:
                G = K.gate_map_cumulative(pooled, G, raw[p + "SA.spatial.weight"], raw[p + "SA.spatial.bias"])
            else:
                x2 = K.spatial_gate(x2, raw[p + "SA.spatial.weight"], raw[p + "SA.spatial.bias"])
            v, part, n = K.qkv_dw(x1, w[p + "qkv_dw"], raw[p + "attn.qkv_dwconv.weight"], gram=True)
            fold = K.mdta_fold(part, n, raw[p + "attn.temperature"], raw[p + "attn.project_out.weight"])
            if self.precision == "fp16x2":
                # as in CVSR_V8: norm2 of the result leaves the 1x1 kernel as fp16 hi | lo planes, and the 3x3 convolution runs as a
                # split-fp16, fp32-grade product (a_hi*w_hi + a_lo*w_hi + a_hi*w_lo) on the LDS-DMA ring kernel
                x1, ln = self._conv(v, fold, res1=x1, res2=x2, res2_scale=G,
                                    ln_out=(raw[p + "norm2.body.weight"], raw[p + "norm2.body.bias"]))
                x1 = K.conv_ring(ln, w[p + "conv_hl"], res1=x1, plane_wrap=8)
                continue
            x1 = self._conv(v, fold, res1=x1, res2=x2, res2_scale=G if lazy_gate else None)
            ln = K.layernorm64(x1, raw[p + "norm2.body.weight"], raw[p + "norm2.body.bias"])
            x1 = self._conv(ln, w[p + "conv"], pad=1, res1=x1, exact=True)
        return x1

    def _rdab(self, w, res, xc, noise, du0=None):
        """``RDAB.forward`` (arch.py:2830-2847).  du0: relu(conv_du_re.0(res)) in space-to-depth form [B,h/2+1,w/2+1,256] when the
        stem kernel already produced it from the residual image (16-bit modes, even sizes); `res` is then unused."""
        raw = w["raw"]
        if du0 is not None:
            t = self._conv(du0, w["RDAB.conv_du_re.2_s2d"], pad=1, act=K.ACT_RELU, exact=True)      # [B, h/2+1, w/2+1, 64]
        else:
            t = self._conv(res, w["RDAB.conv_du_re.0"], act=K.ACT_RELU)
            t = self._conv(t, w["RDAB.conv_du_re.2"], stride=2, pad=2, act=K.ACT_RELU)
        part, n = K.chan_sum_partial(t)
        vmax = K.vec_mlp(part, n, t.shape[1] * t.shape[2], raw["RDAB.conv_du_re2.0.weight"],
                         raw["RDAB.conv_du_re2.0.bias"], 64, K.ACT_RELU)
        xf = self._conv(self._conv(xc, w["RDAB.conv_dc.0"], act=K.ACT_LRELU), w["RDAB.conv_dc.2"])
        mixed = K.rdab_mix(xf, xc, raw["RDAB.spatial.weight"], raw["RDAB.spatial.bias"], vmax, noise)
        return self._conv(mixed, w["RDAB.conv_df.0"], act=K.ACT_LRELU)

    def _block(self, w, p, xs):
        """``Block.forward`` over the level list (arch.py:367-375): x + body(x) + down-exchange + up-exchange, where the
        finest level takes its own body output in place of a down-exchange and the coarsest in place of an up-exchange."""
        b0, b2 = w[p + "body.0"], w[p + "body.2"]
        if self.precision == "fp16x2" and b0.wh is not None and all(z.shape[1] % 4 == 0 for z in xs):
            # the body on CVSR_V8's Block_ kernels: 64->256 weights-stationary, 256->64 on the LDS-DMA ring kernel, fp16
            # chunk-planar tensors in between, single-pass fp16 MFMA with fp32 accumulation
            res = [K.conv_ring(K.conv3x3_body0(K.to_cp16(z), b0, act=K.ACT_LRELU), b2) for z in xs]
        else:
            res = [self._conv(self._conv(z, b0, pad=1, act=K.ACT_LRELU), b2, pad=1) for z in xs]
        outs = []
        last = len(xs) - 1
        for l, (x, r) in enumerate(zip(xs, res)):
            if l == 0:
                o = K.lincomb(x, 1.0, r, 2.0)                                            # d = r
            else:   # down(r[l-1]) = mean2x2(1x1(r[l-1])) = 1x1(mean2x2(r[l-1])); x and r ride on the conv's epilogue
                o = self._conv(K.resample2(res[l - 1], up=False), w[p + "down.0"], res1=x, res2=r)
            if l == last:
                K.lincomb(o, 1.0, r, 1.0, out=o)                                         # u = r
            else:   # up(r[l+1]) = bilinear_x2(1x1(r[l+1])), accumulated into o
                K.resample2(self._conv(res[l + 1], w[p + "up.0"]), up=True, out=o, accumulate=True)
            outs.append(o)
        return outs

    def _trunk(self, w, xs):
        """``SCNet`` / ``SCGroup`` over the level list (arch.py:409-467)."""
        ys = xs
        for g in range(7):
            rs = ys
            for b in range(3):
                rs = self._block(w, f"recon_trunk.body.{g}.body.{b}.", rs)
            ys = [self._conv(r, w[f"recon_trunk.body.{g}.conv"], pad=1, res1=y, res2=(x if g == 6 else None))
                  for x, y, r in zip(xs, ys, rs)]
        return ys

    # -- forward ---------------------------------------------------------------------------------------------------
    def forward(self, x, mvs0, mvs1, pms, rms, ufs, pre_L1_fea=None, gumbel_uniform=None):
        if not x.is_cuda:
            raise NotImplementedError("CVSR_V7 (HIP): CPU tensors are not supported; there is no CPU fallback")
        with K.on_device(x):     # the operands' device becomes the current one: streams, per-device caches of the library
            return self._forward(x, mvs0, mvs1, pms, rms, ufs, pre_L1_fea, gumbel_uniform)

    def capture(self, x, mvs0, mvs1, pms, rms, ufs, pre_L1_fea=None, gumbel_uniform=None):
        """One inference forward at these operands' shapes as a HIP graph (``cdfo_amd.graph.CapturedForward``, as ``CVSR_V8.capture``):
        ~1 000 launches on up to three streams replayed from one ``hipGraphLaunch``.  The model has no fp16 range guard to settle."""
        from .graph import CapturedForward
        return CapturedForward(self, x, mvs0, mvs1, pms, rms, ufs, pre_L1_fea, gumbel_uniform, check_range=False)

    def _forward(self, x, mvs0, mvs1, pms, rms, ufs, pre_L1_fea=None, gumbel_uniform=None):
        B, N, C, H, W = x.shape
        if N != NFRAMES or C != 1:
            raise ValueError(f"expected x of shape [B,7,1,H,W], got {tuple(x.shape)}")
        if H % 4 or W % 4:
            raise ValueError(f"H and W must be multiples of 4 (two pyramid halvings, arch.py:4268-4271); got {H}x{W}")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # training call: the operator graph under autograd, HIP kernels in both directions (cdfo_amd/cvsr_v7_train.py)
            if pre_L1_fea is not None:
                raise NotImplementedError("CVSR_V7 (HIP): the cached-feature path is an inference path; training uses fresh clips")
            for prm in self.parameters():
                if not prm.is_cuda or prm.dtype != torch.float32:
                    raise NotImplementedError("CVSR_V7 (HIP): fp32 parameters on the GPU expected; call .cuda()")
            noise = gumbel_uniform if gumbel_uniform is not None else self.gumbel_uniform
            if noise is not None and len(noise) != N_DRAWS:
                raise ValueError(f"gumbel_uniform must hold the {N_DRAWS} draws of one forward, got {len(noise)}")
            from .cvsr_v7_train import forward_train
            return forward_train(self, x, mvs0, mvs1, pms, rms, ufs, noise)
        w = self._weights()
        raw = w["raw"]
        ctr = self.center
        x = x.contiguous().float()
        pms = pms.contiguous().float()
        mvs = (mvs0.contiguous().float(), mvs1.contiguous().float())
        P = H * W

        # 1. feature extraction (arch.py:4256-4266)
        if pre_L1_fea is None:
            f = K.stem_conv(x, P, B * N, H, W, raw["conv_first.weight"], raw["conv_first.bias"], K.ACT_LRELU)
            s = K.stem_conv(pms, P, B * N, H, W, raw["conv_second.weight"], raw["conv_second.bias"])
            L1 = self._feature_extraction(w, f, s)
        else:
            last_x, last_p = x[:, -1].contiguous(), pms[:, -1].contiguous()
            f = K.stem_conv(last_x, P, B, H, W, raw["conv_first.weight"], raw["conv_first.bias"], K.ACT_LRELU)
            s = K.stem_conv(last_p, P, B, H, W, raw["conv_second.weight"], raw["conv_second.bias"])
            new = self._feature_extraction(w, f, s)
            pre = CVSR_V8._as_pixel_major(pre_L1_fea, B * N, H, W)
            L1 = torch.empty_like(pre)
            L1v, prev_v = L1.view(B, N, H, W, NF), pre.view(B, N, H, W, NF)
            L1v[:, :-1].copy_(prev_v[:, 1:])
            L1v[:, -1].copy_(new)
        # 2. feature pyramid (arch.py:4267-4272), frame-major per level so that one frame of all clips is one tensor
        pyr = [L1, K.resample2(L1, up=False)]
        pyr.append(K.resample2(pyr[1], up=False))
        if ufs.shape[1] != 1:
            ufs, rms = ufs.transpose(1, 2), rms.transpose(1, 2)
        ufs = ufs.contiguous().float()
        rms = rms.contiguous().float()
        noise = gumbel_uniform if gumbel_uniform is not None else self.gumbel_uniform
        if noise is not None and len(noise) != N_DRAWS:
            raise ValueError(f"gumbel_uniform must hold the {N_DRAWS} draws of one forward, got {len(noise)}")
        align = self.MV_deform_align
        align.precision = self.precision
        dev = x.device
        draw = 0
        nstr = int(getattr(self, "neighbour_streams", 0))
        main = torch.cuda.current_stream(dev)
        side = []
        if nstr > 1:
            cache = self.__dict__.setdefault("_side_streams", {})
            side = cache.get((dev, nstr))
            if side is None:
                side = cache[(dev, nstr)] = [torch.cuda.Stream(dev) for _ in range(nstr)]
        prev = None
        fused_pyr: List[torch.Tensor] = []
        # 3. per level, coarse to fine (arch.py:4275-4347)
        for lv in (2, 1, 0):
            h, wd = H >> lv, W >> lv
            Lf = (K.swap_outer(pyr[lv], B, N) if B > 1 else pyr[lv]).view(N, B, h, wd, NF)
            centre_nchw = K.nhwc_to_nchw(Lf[ctr])

            def neighbour(i, mv_all, draw):
                if lv == 0:
                    mv = mv_all[:, i].contiguous()
                    u_img, r_img = ufs[:, :, i], rms[:, :, i]
                    bstride = N * P
                else:
                    mv = K.shrink_planes(mv_all[:, i], lv)
                    u_img, r_img = K.shrink_planes(ufs[:, :, i], lv), K.shrink_planes(rms[:, :, i], lv)
                    bstride = h * wd
                ufs_prior = K.stem_conv(u_img, bstride, B, h, wd, raw["conv_expand_ufs.weight"], raw["conv_expand_ufs.bias"])
                du0 = rms_prior = None
                if self.precision != "f32" and h % 2 == 0 and wd % 2 == 0:
                    # fea_com = fea_i + rms_prior and du0 = relu(conv_du_re.0(rms_prior)) (space-to-depth, + one zero row / column)
                    # from the residual image itself: rms_prior is never written
                    fea_com = K.empty_act(B, h, wd, NF, dev)
                    du0 = K.empty_act(B, h // 2 + 1, wd // 2 + 1, 4 * NF, dev)
                    du0[:, h // 2].zero_()
                    du0[:, :, wd // 2].zero_()
                    K.stem_conv2(r_img, bstride, B, h, wd, raw["conv_expand_rms.weight"], raw["conv_expand_rms.bias"], Lf[i], fea_com,
                                 w["rms_du0"][0], w["rms_du0"][1], K.ACT_RELU, du0, s2dB=True)
                else:
                    rms_prior, fea_com = K.stem_conv(r_img, bstride, B, h, wd, raw["conv_expand_rms.weight"],
                                                     raw["conv_expand_rms.bias"], add=Lf[i])
                if prev is not None:
                    K.resample2(prev[i], up=True, out=fea_com, accumulate=True)
                if noise is None:
                    u = torch.rand((B, NF, h, wd), device=dev, dtype=torch.float32).clamp_min_(1e-30)
                else:
                    u = noise[draw].to(device=dev, dtype=torch.float32).contiguous()
                x_n = self._rdab(w, rms_prior, fea_com, u, du0)
                if self.precision == "fp16x2" and os.environ.get("CDFO_V7_FEAR_1PASS", "1") != "0":
                    # activations rounded once to fp16 (one MFMA pass): out_vs_golden unchanged at 7-10e-5 (developer A/B switch)
                    fea_i = K.conv([Lf[i], x_n], w["conv_expand_fea_r"], pad=1, prec=K.PREC_FP16X1)
                else:
                    fea_i = self._conv([Lf[i], x_n], w["conv_expand_fea_r"], pad=1)
                out = K.nchw_to_nhwc(align.forward_pm(centre_nchw, Lf[ctr], fea_i, ufs_prior, mv))
                return out

            # the twelve neighbour pipelines of a level (backward pass i = 6..0 with mvs0, forward pass i = 0..6 with mvs1)
            # only read the level's features, the previous level's result and the priors: they are issued round-robin on
            # side streams and joined before fb_fusion.  The noise draw index keeps the reference's call order.
            jobs = [(0, i) for i in range(N - 1, -1, -1) if i != ctr] + [(1, i) for i in range(N) if i != ctr]
            for st in side:
                st.wait_stream(main)
            res = {}
            for n, (which, i) in enumerate(jobs):
                ctx = torch.cuda.stream(side[n % len(side)]) if side else contextlib.nullcontext()
                with ctx:
                    res[(which, i)] = neighbour(i, mvs[which], draw + n)
                if side:
                    res[(which, i)].record_stream(main)
            draw += len(jobs)
            for st in side:
                main.wait_stream(st)
            cur = torch.empty((N, B, h, wd, NF), dtype=torch.float32, device=dev)
            for i in range(N):
                if i == ctr:
                    cur[i].copy_(Lf[ctr])
                else:
                    self._conv([res[(0, i)], res[(1, i)]], w["fb_fusion"], out=cur[i])
            prev = cur
            fused = self._conv([cur[i] for i in range(N)], w["tsa_fusion"], act=K.ACT_LRELU)
            if self.debug_taps is not None:
                self.debug_taps[f"fused_L{lv + 1}"] = fused
            fused_pyr.append(fused)
        # 4. cross-scale trunk, 5. pyramid merge + upsampling + skip (arch.py:4352-4366)
        outs = self._trunk(w, fused_pyr[::-1])
        if self.debug_taps is not None:
            self.debug_taps.update(L1_fea=L1, trunk_L1=outs[0])
        o3 = self._conv(outs[2], w["upconv1_L3s"], act=K.ACT_LRELU)                  # [B,h/2,w/2,16]
        o3 = K.conv(o3, w["shuffle_L3"])                                             # [B,H,W,16] (4 real + 12 zero), exact
        o2 = self._conv(outs[1], w["upconv1_L2"], act=K.ACT_LRELU)                   # [B,H,W,16]
        t = self._conv([outs[0], o2, o3], w["upconv1"], act=K.ACT_LRELU)             # [B,2H,2W,64]
        if self.precision != "f32":
            # CVSR_V8's fused tail: upconv2 + LeakyReLU + conv_last's tap sums in one kernel, then the tap gather + x4 skip -- the
            # [B,4H,4W,64] map is never written (V7's conv_last is 1x1: the centre tap of the 3x3 stencil that kernel applies)
            out = K.upconv_last(t, w["upconv2"], w["conv_last3"], raw["conv_last.bias"], x[:, ctr], N * P)
        else:
            t = self._conv(t, w["upconv2"], act=K.ACT_LRELU)                         # [B,4H,4W,64]
            out = K.conv_last(t, w["conv_last3"], raw["conv_last.bias"], x[:, ctr], N * P)
        return out, L1.permute(0, 3, 1, 2)
