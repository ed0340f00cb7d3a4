"""Import-path compatibility with the reference: ``from arch.SIDECVSR_our import CVSR_V8``
(train_LD_37.py:20, train_RA_37.py:20, test_LD_22_FPS.py:8).  The class is the MI355X/HIP implementation."""
from cdfo_amd.cvsr_v8 import CVSR_V8  # noqa: F401

__all__ = ["CVSR_V8"]
