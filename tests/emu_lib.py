"""Loader for the host index model (tests/emu/emu.cpp) -- test infrastructure."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "emu", "emu.cpp")
OUT = os.path.join(HERE, "emu", "libntt_emu.so")
CSRC = os.path.join(os.path.dirname(HERE), "ntt_aie_amd", "csrc")

_lib = None


def lib():
    global _lib
    if _lib is None:
        deps = [SRC] + [os.path.join(CSRC, f) for f in ("pass.h", "field.h", "plan.h")]
        if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", SRC, "-o", OUT])
        L = C.CDLL(OUT)
        L.emu_transform.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint64]
        L.emu_forward_product.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_uint32, C.c_uint64, C.c_uint32]
        L.emu_polymul_fused.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.emu_plan.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.emu_geometry.argtypes = [C.c_int] * 6 + [C.c_uint64, C.c_uint32, C.c_int, C.POINTER(C.c_uint32)]
        for n in ("emu_gl_mul", "emu_gl_add", "emu_gl_sub"):
            getattr(L, n).restype = C.c_uint64
            getattr(L, n).argtypes = [C.c_uint64, C.c_uint64]
        for n in ("emu_m64_mul_plain", "emu_m64_mul", "emu_m64_add", "emu_m64_sub"):
            getattr(L, n).restype = C.c_uint64
            getattr(L, n).argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        for n in ("emu_m32_mul_plain", "emu_m32_add", "emu_m32_sub"):
            getattr(L, n).restype = C.c_uint32
            getattr(L, n).argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        _lib = L
    return _lib


def pack_passes(*ms):
    v = 0
    for i, m in enumerate(ms):
        v |= m << (4 * i)
    return v
