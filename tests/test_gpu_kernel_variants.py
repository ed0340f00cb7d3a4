"""The trunk kernels exist in several forms selected by developer switches that are read ONCE per process:
the default forms are what every other GPU test exercises; this file re-runs the per-kernel parity tests of
tests/test_gpu_conv.py with each switch flipped, in a fresh process per switch, so that the A/B forms stay correct."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VARIANTS = [
    ({"CDFO_WS_RING": "0"}, "ws"),            # Block_.body[0] on the private-halo kernel of rounds 1-2
    ({"CDFO_WS_MFMA16": "0"}, "ws"),          # ring-fed form on v_mfma_f32_32x32x16_f16
    ({"CDFO_RING_SPLIT": "0"}, "ring"),       # four-tap ring convolution on the eight-identical-waves kernel
    ({"CDFO_RING_MFMA16": "0"}, "ring"),      # wave-specialised four-tap form with 32x32x16 consumers
    ({"CDFO_RING_TOUCH": "1"}, "ring"),       # producer-side touches of the tile's residual lines
]


@pytest.mark.gpu
@pytest.mark.parametrize("env_extra,select", VARIANTS, ids=lambda v: "-".join(f"{k}={x}" for k, x in v.items()) if isinstance(v, dict) else v)
def test_kernel_variant_passes_the_per_kernel_parity_tests(env_extra, select):
    from conftest import clean_process_run
    env = dict(os.environ)
    env.update(env_extra)
    cmd = [sys.executable, "-m", "pytest", os.path.join("tests", "test_gpu_conv.py"), "-x", "-q", "-m", "gpu", "-k", select,
           "-p", "no:cacheprovider"]
    rc, out, err = clean_process_run(cmd, env=env, cwd=ROOT, timeout=600)
    assert rc == 0, f"variant {env_extra}: pytest exit {rc}\n{out[-3000:]}\n{err[-2000:]}"
    assert " passed" in out and " failed" not in out, out[-2000:]
