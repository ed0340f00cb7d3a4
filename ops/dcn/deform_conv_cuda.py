"""Top-level module name the reference binds its native operator under: ``ops/dcn/deform_conv.py:4,11`` does
``sys.path.append(os.path.dirname(__file__)); import deform_conv_cuda``.  With this file in ``ops/dcn`` the same two lines
find the MI355X/HIP implementation (the five entry points of ops/dcn/src/deform_conv_cuda.cpp:681-695, backed by
libcdfo_hip.so); ``ops.dcn.deform_conv_cuda`` is the same module under its package path."""
from cdfo_amd.deform_conv_cuda import (deform_conv_backward_input_cuda, deform_conv_backward_parameters_cuda,  # noqa: F401
                                       deform_conv_forward_cuda, modulated_deform_conv_cuda_backward,
                                       modulated_deform_conv_cuda_forward)

__all__ = ["deform_conv_forward_cuda", "deform_conv_backward_input_cuda", "deform_conv_backward_parameters_cuda",
           "modulated_deform_conv_cuda_forward", "modulated_deform_conv_cuda_backward"]
