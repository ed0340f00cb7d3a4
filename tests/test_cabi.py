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
    assert lib.cdfo_sizeof_conv_args() == C.sizeof(_lib.ConvArgs)          # the ctypes mirror and the header agree field by field
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


def test_top_level_deform_conv_cuda_import_like_the_reference():
    """ops/dcn/deform_conv.py:4,11 of the reference: `sys.path.append(dirname(__file__)); import deform_conv_cuda` -- the
    operator module is found as a TOP-LEVEL name from ops/dcn.  Run in a child so this process's sys.path stays as it is."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, os\n"
            f"sys.path.insert(0, {root!r})\n"
            f"sys.path.append(os.path.join({root!r}, 'ops', 'dcn'))\n"
            "import deform_conv_cuda\n"
            "names = ['deform_conv_forward_cuda', 'deform_conv_backward_input_cuda', 'deform_conv_backward_parameters_cuda',\n"
            "         'modulated_deform_conv_cuda_forward', 'modulated_deform_conv_cuda_backward']\n"
            "assert all(callable(getattr(deform_conv_cuda, n)) for n in names)\n"
            "import ops.dcn.deform_conv_cuda as pkg, cdfo_amd.deform_conv_cuda as impl\n"
            "assert all(getattr(pkg, n) is getattr(impl, n) is getattr(deform_conv_cuda, n) for n in names)\n"
            "print('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp")
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr


def test_pack_modules_expose_init_offset():
    """deform_conv.py:252-256, 324-328: `init_offset()` is a public method of both *Pack classes (called by the ctor)."""
    from ops.dcn.deform_conv import DeformConvPack, ModulatedDeformConvPack
    a = DeformConvPack(4, 4, 3, padding=1, deformable_groups=2)
    b = ModulatedDeformConvPack(4, 4, 3, padding=1, deformable_groups=2)
    for m, conv in ((a, a.conv_offset), (b, b.conv_offset_mask)):
        assert float(conv.weight.abs().sum()) == 0.0 and float(conv.bias.abs().sum()) == 0.0
        conv.weight.data.fill_(1.0)
        conv.bias.data.fill_(1.0)
        m.init_offset()
        assert float(conv.weight.abs().sum()) == 0.0 and float(conv.bias.abs().sum()) == 0.0
    assert a.conv_offset.out_channels == 2 * 2 * 9 and b.conv_offset_mask.out_channels == 2 * 3 * 9
