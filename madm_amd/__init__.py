"""madm_amd -- MI355X-native SD-v1-4 single-timestep feature extractor for MADM.

Importing the package loads ``libmadm_hip.so`` (hand-written gfx950 kernels behind the C ABI of
``include/madm_hip.h``); it raises if the library has not been built.
"""
from . import _debug
from . import _lib  # noqa: F401  (fails loudly when the HIP library is missing)

_debug.install()   # MADM_DEBUG_POISON_HBM: poison every uninitialised device allocation (debug harness, off by default)

__all__ = ["_lib"]
