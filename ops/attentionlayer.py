"""Import-path compatibility with the reference's ``ops/attentionlayer.py``: ``DSTA`` is the MI355X/HIP implementation."""
from cdfo_amd.attentionlayer import DSTA  # noqa: F401
