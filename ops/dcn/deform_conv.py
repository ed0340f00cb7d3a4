"""Import-path compatibility with the reference (``from ops.dcn.deform_conv import ModulatedDeformConv``,
ops/attentionlayer.py:8): the classes are the MI355X/HIP implementations in ``cdfo_amd.dcn``."""
from cdfo_amd.dcn import (DeformConv, DeformConvFunction, DeformConvPack, ModulatedDeformConv,  # noqa: F401
                          ModulatedDeformConvFunction, ModulatedDeformConvPack, deform_conv, modulated_deform_conv)
