"""CPU-only checks of the drop-in boundary: the C-ABI library builds, loads, and exports every symbol that
include/cdfo_hip.h declares; argument validation works without touching a GPU; the nn.Module surface carries the
reference's 261 state_dict entries and refuses CPU execution (no silent fallback)."""
import ctypes as C

import pytest
import torch


def test_library_exports_every_declared_symbol():
    from cdfo_amd import _lib
    from cdfo_amd.build import build
    build()
    protos = _lib.header_prototypes()
    assert len(protos) >= 28
    raw = C.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(raw, name), f"{name} declared in include/cdfo_hip.h but not exported"
    lib = _lib.lib()
    assert lib.cdfo_abi_version() == 1
    assert b"gfx950" in lib.cdfo_build_info()


def test_argument_validation_needs_no_gpu():
    from cdfo_amd import _lib
    lib = _lib.lib()
    a = _lib.ConvArgs()
    a.nsrc = 0
    assert lib.cdfo_conv_igemm(C.byref(a), None) == -1                       # CDFO_EINVAL before any HIP call
    assert lib.cdfo_dcn_forward(None, None, None, None, None, None, 1, 6, 8, 8, 4, 3, 3, 1, 1, 1, 1, 1, 1, 4, 1, None, 0, None) == -1
    assert lib.cdfo_layernorm64(None, 63, None, None, 10, None, 64, None) == -1
    assert lib.cdfo_seq_attn(C.c_void_p(8), 64, C.c_void_p(16), 64, C.c_void_p(16), 64, 1, 8, 8, 0, None) == -2  # EALIGN


def test_module_surface_matches_the_reference_state_dict():
    from arch.SIDECVSR_our import CVSR_V8
    from oracle.cvsr_v8_ref import make_state_dict, state_dict_spec
    m = CVSR_V8(SCGs=8)                       # train_LD_37.py:420 passes SCGs=8; accepted and ignored like the reference
    sd = m.state_dict()
    spec = {k: tuple(s) for k, s, *_ in state_dict_spec()}
    assert set(sd) == set(spec) and len(sd) == 261
    for k, v in sd.items():
        assert tuple(v.shape) == spec[k], k
    m.load_state_dict(make_state_dict(3), strict=True)
    assert sum(p.numel() for p in m.parameters()) == 7_098_392


def test_no_cpu_fallback():
    from arch.SIDECVSR_our import CVSR_V8
    from oracle.cvsr_v8_ref import make_inputs
    m = CVSR_V8()
    a = make_inputs(1, 8, 8, 1)
    with torch.no_grad(), pytest.raises(NotImplementedError):
        m(a["x"], a["mvs0"], a["mvs1"], a["pms"], a["rms"], a["ufs"])
    from cdfo_amd.dcn import deform_conv
    with pytest.raises(NotImplementedError):
        deform_conv(torch.zeros(1, 2, 3, 3), torch.zeros(1, 18, 3, 3), torch.ones(1, 2, 3, 3), 1, 1, 1, 1, 1)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from cdfo_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.CdfoError):
        _lib.lib()
