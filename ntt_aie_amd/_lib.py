"""ctypes binding of libntt_hip.so (include/ntt_hip.h).

The product path has no CPU fallback: if the HIP library is missing or fails to
load, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# The package reads NO environment variable (include/ntt_hip.h promises that no variable can change a result; the C
# library itself reads only NTT_ROCTX, which switches profiling ranges).  Experiments that want another build of the
# library say so in code: tools/_explib.py calls use_library(path, allow_experiment=True) before the first transform.
LIB_PATH = os.path.join(_HERE, "libntt_hip.so")
_allow_experiment = False

# error codes of include/ntt_hip.h
NTT_OK = 0
NTT_E_ARG = -1
NTT_E_PRIME = -2
NTT_E_LOGN = -3
NTT_E_NOTABLE = -4
NTT_E_NOTINVERTIBLE = -5
NTT_E_LAYOUT = -6
NTT_E_RANGE = -7
NTT_E_NODEVICE = -8
NTT_E_NOMEM = -9
NTT_E_INTERNAL = -10

LAYOUT_NATURAL = 0
LAYOUT_AIE_BLOCK16 = 1

# every symbol include/ntt_hip.h declares
EXPORTS = (
    "ntt_version", "ntt_error_string", "ntt_device_count", "ntt_plan_create", "ntt_plan_destroy",
    "ntt_plan_set_twiddles", "ntt_make_roots", "ntt_make_table", "ntt_plan_generate_twiddles", "ntt_plan_get_twiddles", "ntt_plan_info", "ntt_plan_select", "ntt_plan_set_policy", "ntt_plan_clone", "ntt_forward",
    "ntt_forward_profile", "ntt_inverse", "ntt_pointwise_mul", "ntt_polymul_negacyclic", "ntt_count_noncanonical", "ntt_forward_stages",
)


def kernel_source_hash() -> str:
    """16 hex digits over the device-code sources (csrc/*.h, *.inc and the kernel *.hip files; not the host-side C-ABI
    ntt_api.hip / guard.h): the identity of the kernels a profile was collected on.  bench.py and tools/*_summary.py stamp it into
    what they write, and bench.py refuses to quote counter values whose stamp differs from the tree it runs in."""
    import glob
    import hashlib

    h = hashlib.sha256()
    src = os.path.join(_HERE, "csrc")
    for f in sorted(glob.glob(os.path.join(src, "*.h")) + glob.glob(os.path.join(src, "*.inc")) + glob.glob(os.path.join(src, "*.hip"))):
        if os.path.basename(f) in ("ntt_api.hip", "guard.h"):  # host side of the C-ABI: no device code
            continue
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def open_library(path: str, since_v3: bool = True) -> C.CDLL:
    """dlopen one build of the C-ABI and declare its signatures (the product library, or tools' experiment build).
    since_v3=False: an older build without the round-3 entry points (tools/regress_sweep.py times one against the tree)."""
    L = C.CDLL(path)
    vp, sz, u64 = C.c_void_p, C.c_size_t, C.c_uint64
    L.ntt_version.restype = C.c_int
    L.ntt_error_string.restype = C.c_char_p
    L.ntt_error_string.argtypes = [C.c_int]
    L.ntt_device_count.restype = C.c_int
    L.ntt_plan_create.argtypes = [C.POINTER(vp), C.c_int, u64, C.c_int, C.c_int]
    L.ntt_plan_destroy.argtypes = [vp]
    L.ntt_plan_set_twiddles.argtypes = [vp, vp]
    L.ntt_make_roots.argtypes = [vp, u64, vp]
    L.ntt_make_table.argtypes = [vp, C.c_int, u64, vp]
    L.ntt_plan_generate_twiddles.argtypes = [vp, C.c_int, u64]
    L.ntt_plan_get_twiddles.argtypes = [vp, C.c_int, vp]
    L.ntt_plan_info.restype = C.c_int64
    L.ntt_plan_info.argtypes = [vp, C.c_int]
    if since_v3:
        L.ntt_plan_select.argtypes = [vp, sz]
        L.ntt_plan_set_policy.argtypes = [vp, C.c_int]
        L.ntt_plan_clone.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.ntt_forward.argtypes = [vp, vp, vp, sz, C.c_int, vp]
    L.ntt_forward_profile.argtypes = [vp, vp, vp, sz, C.c_int, vp, C.POINTER(C.c_float), C.c_int,
                                      C.POINTER(C.c_int)]
    L.ntt_inverse.argtypes = [vp, vp, vp, sz, C.c_int, C.c_int, vp]
    L.ntt_pointwise_mul.argtypes = [vp, vp, vp, vp, sz, u64, vp]
    L.ntt_polymul_negacyclic.argtypes = [vp, vp, vp, vp, sz, vp]
    L.ntt_count_noncanonical.argtypes = [vp, vp, sz, C.POINTER(C.c_uint64)]
    L.ntt_forward_stages.argtypes = [vp, vp, vp, sz, C.c_int, vp]
    return L


class NTTError(RuntimeError):
    def __init__(self, code: int, where: str):
        self.code = code
        msg = lib().ntt_error_string(code)
        super().__init__("%s failed: %d (%s)" % (where, code, msg.decode() if msg else "?"))


_lib = None


def is_experiment_build(L: C.CDLL) -> bool:
    """libntt_hip_exp.so (-DNTT_EXPERIMENT: debug switches that redirect loads / skip stores) exports ntt_plan_set_debug;
    the product library does not."""
    try:
        L.ntt_plan_set_debug
    except AttributeError:
        return False
    return True


def use_library(path: str, allow_experiment: bool = False) -> None:
    """Select another build of the C-ABI for this process (tools/ A/B experiments).  Must run before the first lib() call."""
    global LIB_PATH, _allow_experiment
    if _lib is not None:
        raise RuntimeError("use_library() after the library was loaded (%s)" % LIB_PATH)
    LIB_PATH = os.path.abspath(path)
    _allow_experiment = bool(allow_experiment)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libntt_hip.so is not built (%s). Build it with `make -C ntt_aie_amd/csrc` or "
                "`python -c 'import __graft_entry__ as g; g.build()'`; there is no CPU fallback." % LIB_PATH)
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7.  Import
        # torch first so that this library's NEEDED libamdhip64.so.7 resolves (by soname) to
        # the copy torch already mapped; loading ours first would pull in /opt/rocm's copy and
        # the second runtime to initialise would see no device.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        probe = C.CDLL(LIB_PATH)
        if not hasattr(probe, "ntt_plan_select"):
            # a build from before round 3 (tools/regress_sweep.py times one through open_library(path, since_v3=False)):
            # the package itself needs the launch-time alternatives
            probe.ntt_version.restype = C.c_int
            raise ImportError("%s is an older build of the C-ABI (ntt_version() = %d, no ntt_plan_select): the package needs "
                              "version >= 300; tools that time an old build call _lib.open_library(path, since_v3=False)"
                              % (LIB_PATH, probe.ntt_version()))
        L = open_library(LIB_PATH)
        if is_experiment_build(L) and not _allow_experiment:
            raise ImportError("%s is an experiment build (-DNTT_EXPERIMENT: its debug switches can skip stores); the product "
                              "path refuses it -- tools select it with use_library(path, allow_experiment=True)" % LIB_PATH)
        _lib = L
    return _lib


def check(code: int, where: str) -> None:
    if code != 0:
        raise NTTError(code, where)
