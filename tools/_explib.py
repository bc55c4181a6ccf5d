"""Which build of libntt_hip the tools/ scripts drive.  The product package reads no environment variable, so the choice
is made HERE, in tools-side code: NTT_HIP_LIB=<path> (ab/libntt_NAME.so from tools/ab_build.sh, or the product library),
default = the experiment build libntt_hip_exp.so (`make -C ntt_aie_amd/csrc exp`: NTT_DEBUG_FLAGS, NTT_PLAN_SPLIT, ...).

    import _explib; _explib.select()             # experiment build unless NTT_HIP_LIB says otherwise
    import _explib; _explib.select(default=None) # product library unless NTT_HIP_LIB says otherwise
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
EXP = os.path.join(ROOT, "ntt_aie_amd", "libntt_hip_exp.so")


def select(default=EXP):
    from ntt_aie_amd import _lib

    path = os.environ.get("NTT_HIP_LIB") or default
    if path:
        _lib.use_library(path, allow_experiment=True)
    return os.path.basename(path) if path else "libntt_hip.so"
