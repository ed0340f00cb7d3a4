"""Thin re-export of the oracle's DCN loader (oracle/dcn_modules_ref.py).  Test infrastructure."""
from oracle.dcn_modules_ref import dcn_forward_ref  # noqa: F401
