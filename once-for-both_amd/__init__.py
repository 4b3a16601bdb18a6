"""once-for-both_amd — MI355X-native implementation of the Once-for-Both (OFB) search-training hot path.

Host side: Python modules that mirror the reference's module API.
Device side: hand-written HIP kernels for gfx950 behind the C ABI in include/ofb_hip.h
(csrc/ -> csrc/libofb_hip.so, loaded with ctypes by `hip.py`).  There is no CPU fallback:
every compute entry point raises if the HIP library or a GPU is missing.
"""
from . import hip  # noqa: F401

__all__ = ['hip']
