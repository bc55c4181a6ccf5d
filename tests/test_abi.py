"""The C-ABI library loads and exports every symbol include/ntt_hip.h declares.
No compute calls: this runs on the CPU-only container."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def hiplib():
    import __graft_entry__ as ge

    if not os.path.exists(os.path.join(ROOT, "ntt_aie_amd", "libntt_hip.so")):
        ge.build()
    from ntt_aie_amd import _lib

    return _lib


def test_header_symbols_exported(hiplib):
    hdr = open(os.path.join(ROOT, "include", "ntt_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ntt_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(hiplib.EXPORTS)
    L = hiplib.lib()
    for name in sorted(declared):
        assert getattr(L, name) is not None, name


def test_error_contract_without_compute(hiplib):
    L = hiplib.lib()
    assert L.ntt_version() >= 100
    assert L.ntt_error_string(0) == b"ok"
    for code in range(-10, 0):
        assert L.ntt_error_string(code) not in (None, b"", b"unknown error")
    assert L.ntt_error_string(-11) == b"unknown error"
    assert (hiplib.NTT_E_NOMEM, hiplib.NTT_E_INTERNAL) == (-9, -10)
    h = C.c_void_p()
    # argument errors are reported before any device is touched
    assert L.ntt_plan_create(None, 8, 3329, 4, 0) == hiplib.NTT_E_ARG
    assert L.ntt_plan_create(C.byref(h), 8, 3329, 3, 0) == hiplib.NTT_E_ARG
    assert L.ntt_plan_create(C.byref(h), 0, 3329, 4, 0) == hiplib.NTT_E_LOGN
    assert L.ntt_plan_create(C.byref(h), 29, 3329, 4, 0) == hiplib.NTT_E_LOGN
    assert L.ntt_plan_create(C.byref(h), 8, 3330, 4, 0) == hiplib.NTT_E_PRIME
    assert L.ntt_plan_create(C.byref(h), 8, 1 << 40, 8, 0) == hiplib.NTT_E_PRIME   # even modulus, 8-byte words
    assert L.ntt_plan_create(C.byref(h), 8, 1, 8, 0) == hiplib.NTT_E_PRIME
    assert L.ntt_plan_create(C.byref(h), 8, (1 << 32) + 15, 4, 0) == hiplib.NTT_E_PRIME
    assert L.ntt_forward(None, None, None, 1, 0, None) == hiplib.NTT_E_ARG
    assert L.ntt_plan_destroy(None) == hiplib.NTT_E_ARG
    if L.ntt_device_count() == 0:
        assert L.ntt_plan_create(C.byref(h), 8, 3329, 4, 0) == hiplib.NTT_E_NODEVICE


# entry points that cannot throw and have no error code to return: a constant, a switch over string literals,
# one runtime call returning a count ("never fails": 0 devices on error)
UNGUARDED = {"ntt_version", "ntt_error_string", "ntt_device_count"}


def test_every_entry_point_is_behind_the_exception_wall():
    """include/ntt_hip.h: "nothing throws or aborts".  By parsing the source: every function defined in ntt_api.hip's
    extern "C" block that the header declares is a function-try-block -- its body opens with NTT_GUARD and closes with
    NTT_GUARD_END (csrc/guard.h: std::bad_alloc -> NTT_E_NOMEM, anything else -> NTT_E_INTERNAL) -- and no `new` outside
    std::nothrow, no abort()/exit()/assert() sits in the file."""
    api = open(os.path.join(ROOT, "ntt_aie_amd", "csrc", "ntt_api.hip")).read()
    ext = api[api.index('extern "C" {'):]
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "ntt_hip.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(ntt_[a-z_0-9]+)\s*\(", hdr))
    defs = re.findall(r"^(?:int|int64_t|const char \*)\s*(ntt_[a-z_0-9]+)\(([^{;]*?)\)\s*(NTT_GUARD )?\{", ext, flags=re.M | re.S)
    seen = {}
    for name, _, guard in defs:
        seen[name] = bool(guard)
    assert declared <= set(seen), declared - set(seen)
    for name in sorted(declared):
        if name in UNGUARDED:
            continue
        assert seen[name], "%s is not behind NTT_GUARD" % name
    # every opened guard is closed, once
    assert ext.count("NTT_GUARD {") == ext.count("} NTT_GUARD_END") == sum(seen.values())
    body = re.sub(r"//.*", "", api)
    assert not re.search(r"\b(abort|exit|assert)\s*\(", body)
    assert re.findall(r"\bnew\b(?!\s*\(std::nothrow\))", re.sub(r"#include <new>", "", body)) == []


def test_guard_macros_convert_exceptions(tmp_path):
    """csrc/guard.h on the CPU (tests/cxx/guard_test.cpp, g++): bad_alloc -- thrown, and from a real oversized std::vector --
    comes back as NTT_E_NOMEM (-9); length_error, a thrown int, runtime_error as NTT_E_INTERNAL (-10); values pass through."""
    import subprocess

    exe = str(tmp_path / "guard_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cxx", "guard_test.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ["plain=42", "bad_alloc=-9", "vector=-9", "length=-10", "int=-10", "wide=-10"], out.stdout


def test_product_has_no_oracle_dependency():
    """The product package must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "ntt_aie_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".inc", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle_py" not in text and "ntt_oracle" not in text and "libntt_oracle" not in text, f


def test_tools_and_bench_timed_region_do_not_touch_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import or run anything under oracle/:
    no script under tools/ mentions the oracle modules, and bench.py imports them only inside cpu_baseline()."""
    tools = os.path.join(ROOT, "tools")
    for dirpath, _, files in os.walk(tools):
        for f in files:
            if f.endswith((".py", ".sh", ".hip")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle_py" not in text and "libntt_oracle" not in text and "ntt_oracle" not in text, f
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert bench.count("import oracle_py") == 1
    body = bench[bench.index("def cpu_baseline"):bench.index("def device_copy_rate")]
    assert "import oracle_py" in body


def test_generated_instruction_streams_are_current(tmp_path):
    """csrc/gl_asm.h is generated: the committed file must be what tools/gen_gl_asm.py emits today."""
    import subprocess
    import sys

    out = tmp_path / "gl_asm.h"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_gl_asm.py"), str(out)],
                          stderr=subprocess.DEVNULL)
    assert out.read_text() == open(os.path.join(ROOT, "ntt_aie_amd", "csrc", "gl_asm.h")).read()


def test_product_library_reads_no_environment_and_has_no_experiment_paths(hiplib):
    """Experiment knobs (NTT_DEBUG_FLAGS redirects loads / skips stores, NTT_FUSED, NTT_PLAN_SPLIT, NTT_TARGET_WGS*)
    exist only in the -DNTT_EXPERIMENT build that tools/ load; the product .so must not even contain their names,
    and the experimental fused launch must not be linked into it."""
    blob = open(os.path.join(ROOT, "ntt_aie_amd", "libntt_hip.so"), "rb").read()
    for name in (b"NTT_DEBUG_FLAGS", b"NTT_FUSED", b"NTT_PLAN_SPLIT", b"NTT_TARGET_WGS", b"NTT_ONLY_PASS", b"NTT_PASS_VARIANT", b"fused_gl16"):
        assert name not in blob, name
    assert not os.path.exists(os.path.join(ROOT, "ntt_aie_amd", "csrc", "fused_gl16.hip"))


def test_package_reads_no_ntt_environment_variable(hiplib, tmp_path):
    """include/ntt_hip.h promises that no environment variable can change a result.  True of the library (test above) and of
    the Python package: no source under ntt_aie_amd/*.py touches os.environ / getenv, the only NTT_* name in the C sources of
    the product build is NTT_ROCTX (profiling ranges on/off), and the package refuses an experiment build unless a tool asks
    for it in code (use_library(..., allow_experiment=True))."""
    pkg = os.path.join(ROOT, "ntt_aie_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            text = open(os.path.join(pkg, f)).read()
            assert "os.environ" not in text and "getenv" not in text, f
    api = open(os.path.join(pkg, "csrc", "ntt_api.hip")).read()
    product_side = re.sub(r"#if defined\(NTT_EXPERIMENT\).*?#endif", "", api, flags=re.S)
    assert set(re.findall(r'getenv\("([A-Z_0-9]+)"\)', product_side)) == {"NTT_ROCTX"}
    for f in os.listdir(os.path.join(pkg, "csrc")):
        if f.endswith((".h", ".inc", ".hip")) and f != "ntt_api.hip":
            assert "getenv" not in open(os.path.join(pkg, "csrc", f)).read(), f
    # the product library is not an experiment build; the experiment build (when present) is refused by default
    assert not hiplib.is_experiment_build(hiplib.lib())
    exp = os.path.join(pkg, "libntt_hip_exp.so")
    if os.path.exists(exp):
        import subprocess
        import sys

        code = ("import sys; sys.path.insert(0, %r)\nfrom ntt_aie_amd import _lib\n_lib.use_library(%r)\n"
                "try:\n    _lib.lib()\nexcept ImportError as e:\n    print('REFUSED'); sys.exit(0)\nprint('LOADED')" % (ROOT, exp))
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                             env=dict(os.environ, NTT_HIP_LIB=exp))
        assert "REFUSED" in out.stdout, out.stdout + out.stderr
        # ... and a stray NTT_HIP_LIB in the environment is ignored by the package
        code = ("import sys; sys.path.insert(0, %r)\nfrom ntt_aie_amd import _lib\nprint(_lib.LIB_PATH)" % ROOT)
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                             env=dict(os.environ, NTT_HIP_LIB=exp))
        assert out.stdout.strip().endswith("libntt_hip.so"), out.stdout + out.stderr
