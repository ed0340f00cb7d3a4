"""ctypes loader for the oracle's C restatement of the DCN forward (oracle/dcn_ref.c).  Test infrastructure."""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "_build", "libdcn_ref.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.run(["make", "-s", "-C", os.path.join(_ROOT, "oracle")], check=True)
        _lib = C.CDLL(_SO)
        _lib.dcn_forward_ref.restype = C.c_int
        _lib.dcn_forward_ref.argtypes = [C.c_void_p] * 6 + [C.c_int] * 15
    return _lib


def dcn_forward_ref(x, offset, mask, weight, bias, stride=1, pad=0, dil=1, groups=1, dg=1):
    """numpy fp32 arrays in the reference's layouts -> output [B,Co,Ho,Wo]."""
    x = np.ascontiguousarray(x, np.float32)
    offset = np.ascontiguousarray(offset, np.float32)
    weight = np.ascontiguousarray(weight, np.float32)
    mask = None if mask is None else np.ascontiguousarray(mask, np.float32)
    bias = None if bias is None else np.ascontiguousarray(bias, np.float32)
    B, Cc, H, W = x.shape
    Co, _, kh, kw = weight.shape
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    dh, dw = (dil, dil) if isinstance(dil, int) else dil
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    out = np.empty((B, Co, Ho, Wo), np.float32)
    p = lambda a: None if a is None else a.ctypes.data  # noqa: E731
    rc = lib().dcn_forward_ref(p(x), p(offset), p(mask), p(weight), p(bias), p(out), B, Cc, H, W, Co, kh, kw, sh, sw,
                               ph, pw, dh, dw, groups, dg)
    if rc != 0:
        raise ValueError("dcn_forward_ref rejected the shapes")
    return out
