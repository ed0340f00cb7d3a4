"""hipcc leaves one store-data hazard unpadded on gfx950: a vector instruction that overwrites a data register of a 128-bit buffer
store WITH a scalar offset register within two wait states of it (it produced wrong lanes on the first store of a
unit in the Winograd kernel).  This test compiles the kernels that use such stores to assembly and scans it with
tools/check_store_hazard.py -- a source change that lets the compiler schedule such a write behind the store fails here, on the CPU."""
import os, shutil, subprocess, sys, tempfile
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _sources_with_b128_buffer_stores():
    d = os.path.join(ROOT, "cdfo_amd", "csrc")
    return sorted(f for f in os.listdir(d) if f.endswith(".hip") and "raw_buffer_store_b128" in open(os.path.join(d, f)).read())


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc (cross-compiles without a GPU)")
@pytest.mark.parametrize("src", _sources_with_b128_buffer_stores())
def test_no_unpadded_store_data_hazard(src):
    from check_store_hazard import scan
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        r = subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=fast", "-S", "--cuda-device-only",
                            "-o", out, os.path.join(ROOT, "cdfo_amd", "csrc", src)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        hits = scan(out)
    assert not hits, "\n".join(f"{k}: line {ln}: {st} <- {ins}" for k, ln, st, ins in hits[:10])


def test_scanner_finds_the_pattern():
    from check_store_hazard import _scan_lines
    asm = ["_ZN1kE:", "\tbuffer_store_dwordx4 v[174:177], v178, s[24:27], s2 offen", "\tv_cndmask_b32_e64 v174, 0, 1, s[42:43]"]
    assert len(_scan_lines(asm)) == 1
    asm[2:2] = ["\ts_nop 1"]
    assert _scan_lines(asm) == []
    # no scalar offset register: LLVM pads that form itself
    assert _scan_lines(["_ZN1kE:", "\tbuffer_store_dwordx4 v[4:7], v8, s[24:27], 0 offen", "\tv_mov_b32_e32 v4, 0"]) == []
