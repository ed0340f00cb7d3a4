"""Stand-in for the reference's pybind11 module ``deform_conv_cuda`` (ops/dcn/src/deform_conv_cuda.cpp:681-695): the same
five entry points with the same positional signatures, for a caller that brings the reference's OWN
``ops/dcn/deform_conv.py``.  Each is one call of ``cdfo_amd.dcn.launch_forward`` /
``launch_backward`` (the validated C-ABI launchers over ``cdfo_dcn_forward_dt`` / ``cdfo_dcn_backward_dt``).

Conventions kept from the extension: the caller pre-allocates ``output``; the ``columns`` / ``ones`` scratch tensors are
accepted and ignored (sampling and contraction are fused); gradient tensors arrive zero-filled -- grad_input /
grad_weight / grad_bias are accumulated into, grad_offset / grad_mask are assigned; the DCNv1 functions return 1
(cpp:257, 370, 483); non-contiguous ``input`` / ``weight`` raise (cpp:493-494)."""
from __future__ import annotations

from .dcn import Geometry, launch_backward, launch_forward


def deform_conv_forward_cuda(input, weight, offset, output, columns, ones, kW, kH, dW, dH, padW, padH, dilationW,
                             dilationH, group, deformable_group, im2col_step):
    """cpp:151-156"""
    launch_forward(input, offset, None, weight, None, output,
                   Geometry(kH, kW, dH, dW, padH, padW, dilationH, dilationW, group, deformable_group))
    return 1


def modulated_deform_conv_cuda_forward(input, weight, bias, ones, offset, mask, output, columns, kernel_h, kernel_w,
                                       stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group,
                                       deformable_group, with_bias):
    """cpp:486-492"""
    launch_forward(input, offset, mask, weight, bias if with_bias else None, output,
                   Geometry(kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group,
                            deformable_group))


def deform_conv_backward_input_cuda(input, offset, gradOutput, gradInput, gradOffset, weight, columns, kW, kH, dW, dH,
                                    padW, padH, dilationW, dilationH, group, deformable_group, im2col_step):
    """cpp:260-266"""
    launch_backward(input, offset, None, weight, gradOutput,
                    Geometry(kH, kW, dH, dW, padH, padW, dilationH, dilationW, group, deformable_group),
                    grad_x=gradInput, grad_offset=gradOffset)
    return 1


def deform_conv_backward_parameters_cuda(input, offset, gradOutput, gradWeight, columns, ones, kW, kH, dW, dH, padW,
                                         padH, dilationW, dilationH, group, deformable_group, scale, im2col_step):
    """cpp:373-378: ``gradWeight += scale * dW``.  The forward weight is not an argument of this entry point; the product
    needs only its shape, which ``gradWeight`` has."""
    if gradWeight.dim() != 4 or gradWeight.size(1) * group != input.size(1):
        raise RuntimeError("gradWeight must be [Co, C/groups, kH, kW]")
    launch_backward(input, offset, None, gradWeight, gradOutput,
                    Geometry(kH, kW, dH, dW, padH, padW, dilationH, dilationW, group, deformable_group),
                    grad_weight=gradWeight, scale=scale)
    return 1


def modulated_deform_conv_cuda_backward(input, weight, bias, ones, offset, mask, columns, grad_input, grad_weight,
                                        grad_bias, grad_offset, grad_mask, grad_output, kernel_h, kernel_w, stride_h,
                                        stride_w, pad_h, pad_w, dilation_h, dilation_w, group, deformable_group,
                                        with_bias):
    """cpp:566-573"""
    launch_backward(input, offset, mask, weight, grad_output,
                    Geometry(kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group,
                             deformable_group),
                    grad_x=grad_input, grad_offset=grad_offset, grad_mask=grad_mask, grad_weight=grad_weight,
                    grad_bias=grad_bias if with_bias else None)
