"""Python surface of the deformable-convolution operator, mirroring ops/dcn/deform_conv.py of the reference
(boundary B2): ``DeformConvFunction``, ``ModulatedDeformConvFunction``, ``deform_conv``,
``modulated_deform_conv``, ``DeformConv``, ``DeformConvPack``, ``ModulatedDeformConv``, ``ModulatedDeformConvPack``
with the constructor / forward signatures of deform_conv.py:190-201, 234, 264-275, 305, 311-337 and the same error
behaviour (``NotImplementedError`` for CPU tensors, deform_conv.py:46-47, 136-137; ``ValueError`` for non-4D input,
:26-29).  The arithmetic runs in libcdfo_hip.so (``cdfo_dcn_forward`` / ``cdfo_dcn_backward``); ``backward`` follows
deform_conv.py:60-99 and :150-172 (zero-filled gradient tensors handed to the extension's backward entry points)."""
from __future__ import annotations

import math

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.nn.modules.utils import _pair

from . import deform_conv_cuda


class DeformConvFunction(Function):
    @staticmethod
    def forward(ctx, input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1,
                im2col_step=64):
        if input is not None and input.dim() != 4:
            raise ValueError("Expected 4D tensor as input, got {}D tensor instead.".format(input.dim()))
        stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
        if not input.is_cuda:
            raise NotImplementedError
        output = input.new_empty(DeformConvFunction._output_size(input, weight, padding, dilation, stride))
        cur_im2col_step = min(im2col_step, input.shape[0])
        assert (input.shape[0] % cur_im2col_step) == 0, 'im2col step must divide batchsize'
        ctx.stride, ctx.padding, ctx.dilation = stride, padding, dilation
        ctx.groups, ctx.deformable_groups, ctx.im2col_step = groups, deformable_groups, im2col_step
        ctx.save_for_backward(input, offset, weight)
        bufs = [input.new_empty(0), input.new_empty(0)]
        ctx.bufs_ = bufs
        deform_conv_cuda.deform_conv_forward_cuda(
            input.contiguous(), weight.contiguous(), offset, output, bufs[0], bufs[1], weight.size(3), weight.size(2),
            stride[1], stride[0], padding[1], padding[0], dilation[1], dilation[0], groups, deformable_groups,
            cur_im2col_step)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        input, offset, weight = ctx.saved_tensors
        grad_input = grad_offset = grad_weight = None
        if not grad_output.is_cuda:
            raise NotImplementedError
        cur_im2col_step = min(ctx.im2col_step, input.shape[0])
        assert (input.shape[0] % cur_im2col_step) == 0, 'im2col step must divide batchsize'
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            grad_input = torch.zeros_like(input)
            grad_offset = torch.zeros_like(offset)
            deform_conv_cuda.deform_conv_backward_input_cuda(
                input, offset, grad_output, grad_input, grad_offset, weight, ctx.bufs_[0], weight.size(3),
                weight.size(2), ctx.stride[1], ctx.stride[0], ctx.padding[1], ctx.padding[0], ctx.dilation[1],
                ctx.dilation[0], ctx.groups, ctx.deformable_groups, cur_im2col_step)
        if ctx.needs_input_grad[2]:
            grad_weight = torch.zeros_like(weight)
            deform_conv_cuda.deform_conv_backward_parameters_cuda(
                input, offset, grad_output, grad_weight, ctx.bufs_[0], ctx.bufs_[1], weight.size(3), weight.size(2),
                ctx.stride[1], ctx.stride[0], ctx.padding[1], ctx.padding[0], ctx.dilation[1], ctx.dilation[0],
                ctx.groups, ctx.deformable_groups, 1, cur_im2col_step)
        return (grad_input, grad_offset, grad_weight, None, None, None, None, None, None)

    @staticmethod
    def _output_size(input, weight, padding, dilation, stride):
        channels = weight.size(0)
        output_size = (input.size(0), channels)
        for d in range(input.dim() - 2):
            in_size = input.size(d + 2)
            kernel = dilation[d] * (weight.size(d + 2) - 1) + 1
            output_size += ((in_size + (2 * padding[d]) - kernel) // stride[d] + 1,)
        if not all(map(lambda s: s > 0, output_size)):
            raise ValueError("convolution input is too small (output would be {})".format(
                'x'.join(map(str, output_size))))
        return output_size


class ModulatedDeformConvFunction(Function):
    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
                deformable_groups=1):
        with_bias = bias is not None
        if not with_bias:
            bias = input.new_empty(1)  # fake tensor, as in the reference
        if not input.is_cuda:
            raise NotImplementedError
        n, _, height, width = input.shape
        channels_out, _, kernel_h, kernel_w = weight.shape
        height_out = (height + 2 * padding - (dilation * (kernel_h - 1) + 1)) // stride + 1
        width_out = (width + 2 * padding - (dilation * (kernel_w - 1) + 1)) // stride + 1
        output = input.new_empty((n, channels_out, height_out, width_out))
        ctx.stride, ctx.padding, ctx.dilation = stride, padding, dilation
        ctx.groups, ctx.deformable_groups, ctx.with_bias = groups, deformable_groups, with_bias
        if weight.requires_grad or mask.requires_grad or offset.requires_grad or input.requires_grad:
            ctx.save_for_backward(input, offset, mask, weight, bias)
        bufs = [input.new_empty(0), input.new_empty(0)]
        ctx._bufs = bufs
        deform_conv_cuda.modulated_deform_conv_cuda_forward(
            input.contiguous(), weight.contiguous(), bias, bufs[0], offset, mask, output, bufs[1], kernel_h, kernel_w,
            stride, stride, padding, padding, dilation, dilation, groups, deformable_groups, with_bias)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        if not grad_output.is_cuda:
            raise NotImplementedError
        input, offset, mask, weight, bias = ctx.saved_tensors
        grad_input = torch.zeros_like(input)
        grad_offset = torch.zeros_like(offset)
        grad_mask = torch.zeros_like(mask)
        grad_weight = torch.zeros_like(weight)
        grad_bias = torch.zeros_like(bias)
        deform_conv_cuda.modulated_deform_conv_cuda_backward(
            input, weight, bias, ctx._bufs[0], offset, mask, ctx._bufs[1], grad_input, grad_weight, grad_bias,
            grad_offset, grad_mask, grad_output, weight.shape[2], weight.shape[3], ctx.stride, ctx.stride, ctx.padding,
            ctx.padding, ctx.dilation, ctx.dilation, ctx.groups, ctx.deformable_groups, ctx.with_bias)
        if not ctx.with_bias:
            grad_bias = None
        return (grad_input, grad_offset, grad_mask, grad_weight, grad_bias, None, None, None, None, None)


deform_conv = DeformConvFunction.apply
modulated_deform_conv = ModulatedDeformConvFunction.apply


class DeformConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=False):
        super().__init__()
        assert not bias
        assert in_channels % groups == 0, f'in_channels {in_channels} cannot be divisible by groups {groups}'
        assert out_channels % groups == 0, f'out_channels {out_channels} cannot be divisible by groups {groups}'
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.groups, self.deformable_groups = groups, deformable_groups
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *self.kernel_size))
        self.reset_parameters()

    def reset_parameters(self):
        n = self.in_channels
        for k in self.kernel_size:
            n *= k
        stdv = 1. / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)

    def forward(self, x, offset):
        return deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups,
                           self.deformable_groups)


class DeformConvPack(DeformConv):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.conv_offset = nn.Conv2d(self.in_channels,
                                     self.deformable_groups * 2 * self.kernel_size[0] * self.kernel_size[1],
                                     kernel_size=self.kernel_size, stride=_pair(self.stride),
                                     padding=_pair(self.padding), bias=True)
        self.init_offset()

    def init_offset(self):      # public in the reference (deform_conv.py:254-256)
        self.conv_offset.weight.data.zero_()
        self.conv_offset.bias.data.zero_()

    def forward(self, x):
        offset = self.conv_offset(x)
        return deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups,
                           self.deformable_groups)


class ModulatedDeformConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.groups, self.deformable_groups, self.with_bias = groups, deformable_groups, bias
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *self.kernel_size))
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        n = self.in_channels
        for k in self.kernel_size:
            n *= k
        stdv = 1. / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.zero_()

    def forward(self, x, offset, mask):
        return modulated_deform_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding,
                                     self.dilation, self.groups, self.deformable_groups)


class ModulatedDeformConvPack(ModulatedDeformConv):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.conv_offset_mask = nn.Conv2d(self.in_channels,
                                          self.deformable_groups * 3 * self.kernel_size[0] * self.kernel_size[1],
                                          kernel_size=self.kernel_size, stride=_pair(self.stride),
                                          padding=_pair(self.padding), bias=True)
        self.init_offset()

    def init_offset(self):      # public in the reference (deform_conv.py:326-328)
        self.conv_offset_mask.weight.data.zero_()
        self.conv_offset_mask.bias.data.zero_()

    def forward(self, x):
        out = self.conv_offset_mask(x)
        o1, o2, mask = torch.chunk(out, 3, dim=1)
        offset = torch.cat((o1, o2), dim=1)
        mask = torch.sigmoid(mask)
        return modulated_deform_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding,
                                     self.dilation, self.groups, self.deformable_groups)
