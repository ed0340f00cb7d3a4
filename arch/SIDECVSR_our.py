"""Import-path compatibility with the reference: ``from arch.SIDECVSR_our import CVSR_V8``
(train_LD_37.py:20, train_RA_37.py:20, test_LD_22_FPS.py:8); ``CVSR_V7`` (arch.py:4215) and its alignment module
``MVDualAttAlignment`` (arch.py:3265) live at the same path.  The classes are the MI355X/HIP implementations."""
from cdfo_amd.cvsr_v7 import CVSR_V7  # noqa: F401
from cdfo_amd.cvsr_v8 import CVSR_V8  # noqa: F401
from cdfo_amd.mv_align import MVDualAttAlignment  # noqa: F401

__all__ = ["CVSR_V7", "CVSR_V8", "MVDualAttAlignment"]
