"""ctypes/numpy doorway onto oracle/libntt_oracle.so (and oracle/_ref when built).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from the product package ntt_aie_amd.
Every function cites the reference lines its C body follows (see ntt_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libntt_oracle.so")
_REF_PATH = os.path.join(_HERE, "_ref", "libntt_ref.so")

GOLDILOCKS = 0xFFFFFFFF00000001
ANS_ORDER = (0, 2, 1, 3, 8, 10, 9, 11, 4, 6, 5, 7, 12, 14, 13, 15)  # test.cpp:70-71 (data)


def build(force: bool = False) -> None:
    """Compile the C restatement (and the literal reference when its tree is here)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "ntt_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "libntt_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src") and (force or not os.path.exists(_REF_PATH)):
        subprocess.check_call(["bash", os.path.join(_HERE, "build_ref.sh")], stdout=subprocess.DEVNULL)


_lib = None
_ref = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        u32p, u64p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
        L.oracle_modpow.restype = C.c_uint64
        L.oracle_modpow.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.oracle_make_roots_u32.argtypes = [C.c_uint32, u32p, C.c_uint32, C.c_uint32]
        L.oracle_make_roots_u64.argtypes = [C.c_uint64, u64p, C.c_uint64, C.c_uint64]
        L.oracle_ntt_u32.argtypes = [u32p, C.c_uint32, u32p, C.c_uint32, C.c_int]
        L.oracle_ntt_u64.argtypes = [u64p, C.c_uint64, u64p, C.c_uint64, C.c_int]
        L.oracle_intt_u32.argtypes = [u32p, C.c_uint32, u32p, C.c_uint32]
        L.oracle_intt_u64.argtypes = [u64p, C.c_uint64, u64p, C.c_uint64]
        L.oracle_ntt_batch_u32.argtypes = [u32p, C.c_uint32, C.c_size_t, u32p, C.c_uint32, C.c_int]
        L.oracle_ntt_batch_u64.argtypes = [u64p, C.c_uint64, C.c_size_t, u64p, C.c_uint64, C.c_int]
        L.oracle_intt_batch_u32.argtypes = [u32p, C.c_uint32, C.c_size_t, u32p, C.c_uint32, C.c_int]
        L.oracle_intt_batch_u64.argtypes = [u64p, C.c_uint64, C.c_size_t, u64p, C.c_uint64, C.c_int]
        L.oracle_pointwise_u32.argtypes = [u32p, u32p, u32p, C.c_size_t, C.c_uint32, C.c_uint32]
        L.oracle_pointwise_u64.argtypes = [u64p, u64p, u64p, C.c_size_t, C.c_uint64, C.c_uint64]
        L.oracle_block16_u32.argtypes = [u32p, u32p, C.c_uint32]
        L.oracle_block16_u64.argtypes = [u64p, u64p, C.c_uint64]
        L.oracle_make_table_u64.argtypes = [C.c_int, C.c_uint64, u64p, C.c_uint64, C.c_uint64]
        L.oracle_negacyclic_schoolbook_u64.argtypes = [u64p, u64p, u64p, C.c_uint64, C.c_uint64]
        L.oracle_fnv1a64.restype = C.c_uint64
        L.oracle_fnv1a64.argtypes = [C.c_void_p, C.c_size_t]
        _lib = L
    return _lib


def have_ref() -> bool:
    if not os.path.exists(_REF_PATH) and os.path.isdir("/root/reference/src"):
        build()
    return os.path.exists(_REF_PATH)


def ref() -> C.CDLL:
    """The literal reference lines (test.cpp:15-60 etc.), int32 exactly as shipped."""
    global _ref
    if _ref is None:
        if not have_ref():
            raise RuntimeError("oracle/_ref/libntt_ref.so not built (reference tree absent)")
        R = C.CDLL(_REF_PATH)
        i32p = C.POINTER(C.c_int32)
        R.ref_modPow.restype = C.c_int32
        R.ref_modPow.argtypes = [C.c_int32] * 3
        R.ref_make_roots.argtypes = [C.c_int32, i32p, C.c_int32, C.c_int32]
        R.ref_ntt.argtypes = [i32p, C.c_int32, i32p, C.c_int32, C.c_int32]
        R.ref_block_order.argtypes = [i32p, i32p, C.c_int32]
        for name in ("ref_modadd", "ref_modsub"):
            getattr(R, name).restype = C.c_int32
            getattr(R, name).argtypes = [C.c_int32] * 3
        R.ref_barrett_2k.restype = C.c_int32
        R.ref_barrett_2k.argtypes = [C.c_int32] * 5
        _ref = R
    return _ref


def _ptr(a: np.ndarray):
    ct = {np.dtype(np.uint32): C.c_uint32, np.dtype(np.uint64): C.c_uint64,
          np.dtype(np.int32): C.c_int32}[a.dtype]
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(ct))


def _dt(word_bytes: int):
    return np.uint32 if word_bytes == 4 else np.uint64


# ---- restatement ------------------------------------------------------------
def make_roots(n: int, p: int, g: int, word_bytes: int) -> np.ndarray:
    """test.cpp:27-32 + :138."""
    T = np.empty(n, dtype=_dt(word_bytes))
    if word_bytes == 4:
        lib().oracle_make_roots_u32(n, _ptr(T), p, g)
    else:
        lib().oracle_make_roots_u64(n, _ptr(T), p, g)
    return T


def make_table(kind: int, n: int, p: int, g: int, word_bytes: int = 8) -> np.ndarray:
    T = np.empty(n, dtype=np.uint64)
    rc = lib().oracle_make_table_u64(kind, n, _ptr(T), p, g)
    if rc != 0:
        raise ValueError("table kind %d not available for n=%d p=%d" % (kind, n, p))
    return T.astype(_dt(word_bytes))


def ntt(a: np.ndarray, T: np.ndarray, p: int, stage: int | None = None, nthreads: int = 1) -> np.ndarray:
    """test.cpp:34-60 on every row of a ([n] or [batch][n]); returns a new array."""
    out = np.ascontiguousarray(a).copy()
    n = out.shape[-1]
    batch = out.size // n
    T = np.ascontiguousarray(T)
    assert T.dtype == out.dtype and T.shape[0] == n
    if stage is not None:
        assert batch == 1
        (lib().oracle_ntt_u32 if out.dtype == np.uint32 else lib().oracle_ntt_u64)(
            _ptr(out), n, _ptr(T), p, stage)
        return out
    (lib().oracle_ntt_batch_u32 if out.dtype == np.uint32 else lib().oracle_ntt_batch_u64)(
        _ptr(out), n, batch, _ptr(T), p, nthreads)
    return out


def intt(a: np.ndarray, T: np.ndarray, p: int, nthreads: int = 1) -> np.ndarray:
    out = np.ascontiguousarray(a).copy()
    n = out.shape[-1]
    batch = out.size // n
    T = np.ascontiguousarray(T)
    rc = (lib().oracle_intt_batch_u32 if out.dtype == np.uint32 else lib().oracle_intt_batch_u64)(
        _ptr(out), n, batch, _ptr(T), p, nthreads)
    if rc != 0:
        raise ValueError("table has a non-invertible twiddle")
    return out


def pointwise(a: np.ndarray, b: np.ndarray, p: int, scale: int = 1) -> np.ndarray:
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    c = np.empty_like(a)
    (lib().oracle_pointwise_u32 if a.dtype == np.uint32 else lib().oracle_pointwise_u64)(
        _ptr(c), _ptr(a), _ptr(b), a.size, p, scale)
    return c


def block16(a: np.ndarray) -> np.ndarray:
    """test.cpp:212-219 on every row."""
    a = np.ascontiguousarray(a)
    n = a.shape[-1]
    flat = a.reshape(-1, n)
    out = np.empty_like(flat)
    f = lib().oracle_block16_u32 if a.dtype == np.uint32 else lib().oracle_block16_u64
    for r in range(flat.shape[0]):
        f(_ptr(out[r]), _ptr(flat[r]), n)
    return out.reshape(a.shape)


def negacyclic_schoolbook(a: np.ndarray, b: np.ndarray, p: int) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    c = np.empty_like(a)
    lib().oracle_negacyclic_schoolbook_u64(_ptr(c), _ptr(a), _ptr(b), a.shape[0], p)
    return c


def fnv1a64(a: np.ndarray) -> int:
    a = np.ascontiguousarray(a)
    return int(lib().oracle_fnv1a64(a.ctypes.data, a.nbytes))


def modpow(x: int, e: int, p: int) -> int:
    return int(lib().oracle_modpow(x, e, p))


# ---- literal reference (int32, p <= 46340) -----------------------------------
def ref_make_roots(n: int, p: int, g: int) -> np.ndarray:
    T = np.empty(n, dtype=np.int32)
    ref().ref_make_roots(n, _ptr(T), p, g)
    return T


def ref_ntt(a: np.ndarray, T: np.ndarray, p: int, stage: int) -> np.ndarray:
    out = np.ascontiguousarray(a, dtype=np.int32).copy()
    T = np.ascontiguousarray(T, dtype=np.int32)
    ref().ref_ntt(_ptr(out), out.shape[0], _ptr(T), p, stage)
    return out


def ref_block_order(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.int32)
    out = np.empty_like(a)
    ref().ref_block_order(_ptr(out), _ptr(a), a.shape[0])
    return out
