"""cdfo_amd -- MI355X-native (gfx950 HIP) implementation of the CDFO CVSR_V8 forward hot path."""
__version__ = "0.1.0"
