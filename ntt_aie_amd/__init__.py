"""MI355X-native NTT engine: hand-written HIP (gfx950) butterfly-network kernels
behind a C-ABI (include/ntt_hip.h), driven by a thin Python host that keeps the
reference's (input, root, output) buffer contract (hal-lab-u-tokyo/ntt-aie,
src/test.cpp:115-190, src/aie2.py:320-337)."""
from ._lib import (LAYOUT_AIE_BLOCK16, LAYOUT_NATURAL, LIB_PATH, NTTError)  # noqa: F401

__all__ = ["NTTPlan", "MultiDevicePlan", "NTTError", "LAYOUT_NATURAL", "LAYOUT_AIE_BLOCK16", "GOLDILOCKS", "to_device",
           "to_host", "version"]


def version() -> int:
    from . import _lib
    return _lib.lib().ntt_version()


def __getattr__(name):
    if name in ("NTTPlan", "GOLDILOCKS", "to_device", "to_host"):
        from . import plan
        return getattr(plan, name)
    if name == "MultiDevicePlan":
        from . import multi
        return multi.MultiDevicePlan
    raise AttributeError(name)
