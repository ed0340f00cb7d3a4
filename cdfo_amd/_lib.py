"""ctypes binding of libcdfo_hip.so (declared in include/cdfo_hip.h).

There is no CPU fallback: if the library is missing this module raises at import time of the first call, and
every entry point's non-zero status becomes a RuntimeError."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcdfo_hip.so")

_lib = None


class CdfoError(RuntimeError):
    pass


class ConvArgs(C.Structure):
    _fields_ = [
        ("src", C.c_void_p * 3), ("ld", C.c_int * 3), ("cs", C.c_int * 3), ("nsrc", C.c_int),
        ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int),
        ("ks", C.c_int), ("stride", C.c_int), ("pad", C.c_int),
        ("Cin", C.c_int), ("Cout", C.c_int), ("CoutP", C.c_int),
        ("w", C.c_void_p), ("w_bstride", C.c_longlong), ("bias", C.c_void_p),
        ("act", C.c_int),
        ("res1", C.c_void_p), ("ldr1", C.c_int),
        ("res2", C.c_void_p), ("ldr2", C.c_int),
        ("out", C.c_void_p), ("ldo", C.c_int), ("store_mode", C.c_int),
        ("prec", C.c_int),
    ]


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CdfoError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -m cdfo_amd.build`).  There is no CPU fallback for the product path.")
        _lib = C.CDLL(LIB_PATH)
        _lib.cdfo_build_info.restype = C.c_char_p
    return _lib


def check(status: int, what: str) -> None:
    if status != 0:
        kind = {-1: "invalid argument", -2: "misaligned pointer/pitch"}.get(status, f"hipError_t {status}")
        raise CdfoError(f"{what} failed: {kind}")
