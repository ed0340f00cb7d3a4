"""ctypes binding of libcdfo_hip.so (declared in include/cdfo_hip.h).

There is no CPU fallback: if the library is missing this module raises at import time of the first call, and
every entry point's non-zero status becomes a RuntimeError."""
from __future__ import annotations

import ctypes as C
import os

# torch first: its wheel bundles the HIP/HSA runtime (libamdhip64.so.7).  libcdfo_hip.so must bind to THAT copy --
# the streams and device pointers handed across the C-ABI come from it -- so it has to be in the process before
# our library's DT_NEEDED entries are resolved.  (Loading /opt/rocm's copy first leaves two HSA runtimes in one
# process and every launch fails with hipErrorNoDevice.)
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CDFO_LIB_PATH") or os.path.join(_HERE, "lib", "libcdfo_hip.so")   # (env: developer A/B builds)

_lib = None


class CdfoError(RuntimeError):
    pass


class ConvArgs(C.Structure):
    _fields_ = [
        ("src", C.c_void_p * 8), ("ld", C.c_int * 8), ("cs", C.c_int * 8), ("nsrc", C.c_int),
        ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int),
        ("ks", C.c_int), ("stride", C.c_int), ("pad", C.c_int),
        ("Cin", C.c_int), ("Cout", C.c_int), ("CoutP", C.c_int),
        ("w", C.c_void_p), ("w_bstride", C.c_longlong), ("bias", C.c_void_p),
        ("act", C.c_int),
        ("res1", C.c_void_p), ("ldr1", C.c_int),
        ("res2", C.c_void_p), ("ldr2", C.c_int),
        ("out", C.c_void_p), ("ldo", C.c_int), ("store_mode", C.c_int),
        ("prec", C.c_int),
        ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p),
        ("tap_mask", C.c_void_p),
        ("src_f16", C.c_int), ("out_f16", C.c_int),
        ("out2_cp16", C.c_void_p),
        ("src_plane_wrap", C.c_int),
        ("res_up2", C.c_void_p), ("ldru", C.c_int),
        ("out2_lo", C.c_int),
        ("mask_out", C.c_void_p), ("flow", C.c_void_p), ("flow_bstride", C.c_longlong), ("off_mag", C.c_float),
        ("off_accumulate", C.c_int),
        ("res2_pixscale", C.c_void_p),
        ("src_halfsplit", C.c_int),
    ]


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CdfoError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -m cdfo_amd.build`).  There is no CPU fallback for the product path.")
        _lib = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in header_prototypes().items():
            fn = getattr(_lib, name)          # AttributeError here = header/library mismatch: fail loudly
            fn.restype = restype
            fn.argtypes = argtypes
    return _lib


HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "cdfo_hip.h")


def header_prototypes(path: str = HEADER_PATH):
    """Parse include/cdfo_hip.h -> {symbol: (restype, [argtypes])}; the header is the single source of truth."""
    import re
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const char\*|long long|int)\s+(cdfo_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        at = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    at.append(C.c_void_p)
                elif a.startswith("long long"):
                    at.append(C.c_longlong)
                elif a.startswith("float"):
                    at.append(C.c_float)
                elif a.startswith("int"):
                    at.append(C.c_int)
                else:
                    raise CdfoError(f"cannot map C type in prototype of {name}: {a!r}")
        protos[name] = (C.c_char_p if ret.startswith("const char") else C.c_longlong if ret == "long long" else C.c_int, at)
    return protos


def check(status: int, what: str) -> None:
    if status != 0:
        kind = {-1: "invalid argument", -2: "misaligned pointer/pitch"}.get(status, f"hipError_t {status}")
        raise CdfoError(f"{what} failed: {kind}")
