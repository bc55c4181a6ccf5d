"""Python host of the MI355X NTT engine.

Mirrors the reference host procedure (src/test.cpp:115-190): make the twiddle
table with the reference's rule, hand over (input, root, output) buffers, launch,
wait.  torch is used only for device memory and streams; every transform goes
through the C-ABI in libntt_hip.so (ntt_aie_amd/_lib.py).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import LAYOUT_AIE_BLOCK16, LAYOUT_NATURAL, check

GOLDILOCKS = 0xFFFFFFFF00000001


def _np_dtype(word_bytes: int):
    return np.uint32 if word_bytes == 4 else np.uint64


def _torch_dtype(word_bytes: int):
    # torch tensors are plain device memory here: signed types carry the same bits
    return torch.int32 if word_bytes == 4 else torch.int64


def to_device(a: np.ndarray, device) -> torch.Tensor:
    """Host words -> device buffer with the same bit pattern."""
    a = np.ascontiguousarray(a)
    signed = a.view(np.int32 if a.dtype.itemsize == 4 else np.int64)
    return torch.from_numpy(signed).to(device)


def to_host(t: torch.Tensor) -> np.ndarray:
    a = t.detach().cpu().contiguous().numpy()
    return a.view(np.uint32 if a.dtype.itemsize == 4 else np.uint64)


class NTTPlan:
    """One (logn, p, word size) transform plan on one GPU.

    The plan owns the device copies of the twiddle table ("root" buffer,
    src/test.cpp:119-120) and of its inverse; data buffers stay caller-owned.
    """

    def __init__(self, logn: int, p: int, word_bytes: int | None = None, device: int | None = None):
        if word_bytes is None:
            word_bytes = 8 if p >= (1 << 32) else 4
        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        self.logn, self.n, self.p, self.word_bytes, self.device = logn, 1 << logn, p, word_bytes, device
        self._h = C.c_void_p()
        check(_lib.lib().ntt_plan_create(C.byref(self._h), logn, p, word_bytes, device), "ntt_plan_create")
        self.table: np.ndarray | None = None

    def close(self) -> None:
        if self._h:
            _lib.lib().ntt_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- tables (host side, like the reference's make_roots) -----------------
    def make_roots(self, g: int) -> np.ndarray:
        """src/test.cpp:27-32 + :138: T[0]=1, T[i]=T[i-1]*g^((p-1)/N) mod p."""
        return self.make_table(0, g)

    def make_table(self, kind: int, g: int) -> np.ndarray:
        T = np.empty(self.n, dtype=_np_dtype(self.word_bytes))
        check(_lib.lib().ntt_make_table(self._h, kind, g, T.ctypes.data), "ntt_make_table")
        return T

    def set_twiddles(self, T: np.ndarray) -> None:
        T = np.ascontiguousarray(T, dtype=_np_dtype(self.word_bytes))
        if T.shape != (self.n,):
            raise ValueError("twiddle table must have N = %d words" % self.n)
        check(_lib.lib().ntt_plan_set_twiddles(self._h, T.ctypes.data), "ntt_plan_set_twiddles")
        self.table = T

    def generate_twiddles(self, kind: int, g: int) -> None:
        """Make the table (and its inverse) on the device: no host table, no upload."""
        check(_lib.lib().ntt_plan_generate_twiddles(self._h, kind, g), "ntt_plan_generate_twiddles")
        self.table = None

    def get_twiddles(self, inverse: bool = False) -> np.ndarray:
        T = np.empty(self.n, dtype=_np_dtype(self.word_bytes))
        check(_lib.lib().ntt_plan_get_twiddles(self._h, int(inverse), T.ctypes.data), "ntt_plan_get_twiddles")
        return T

    @property
    def hbm_passes(self) -> int:
        return int(_lib.lib().ntt_plan_info(self._h, 3))

    @property
    def passes(self) -> list[tuple[str, int, int]]:
        """[(kind, first stage, stages)] of one forward transform; kind is 'contig' for pass 0, 'col' after."""
        L = _lib.lib()
        return [("contig" if i == 0 else "col", int(L.ntt_plan_info(self._h, 64 + i)), int(L.ntt_plan_info(self._h, 32 + i)))
                for i in range(self.hbm_passes)]

    def passes_for(self, batch: int) -> list[tuple[str, int, int]]:
        """The decomposition ntt_forward / ntt_inverse run for THIS batch (the plan keeps alternatives and the launcher picks
        by batch size and modulus class; `passes` is the default one)."""
        L = _lib.lib()
        alt = int(L.ntt_plan_select(self._h, batch))
        k = int(L.ntt_plan_info(self._h, 256 + 16 * alt))
        return [("contig" if i == 0 else "col", int(L.ntt_plan_info(self._h, 256 + 16 * alt + 8 + i)),
                 int(L.ntt_plan_info(self._h, 256 + 16 * alt + 1 + i))) for i in range(k)]

    @property
    def alternatives(self) -> list[tuple[list[int], int]]:
        """[(stages per pass, smallest batch it is chosen for)] of every launch-time decomposition of this plan."""
        L = _lib.lib()
        out = []
        for a in range(int(L.ntt_plan_info(self._h, 6))):
            k = int(L.ntt_plan_info(self._h, 256 + 16 * a))
            out.append(([int(L.ntt_plan_info(self._h, 256 + 16 * a + 1 + i)) for i in range(k)],
                        int(L.ntt_plan_info(self._h, 256 + 16 * a + 15))))
        return out

    @property
    def alternative_variants(self) -> list[list[int]]:
        """[kernel variant per pass] of every alternative (ntt_plan_info 512+: 0 = the default kernel of the pass shape)."""
        L = _lib.lib()
        return [[int(L.ntt_plan_info(self._h, 512 + 16 * a + i)) for i in range(len(stages))]
                for a, (stages, _) in enumerate(self.alternatives)]

    def set_policy(self, alternative: int) -> None:
        """-1 (default): the launcher picks the decomposition by batch; k >= 0: always alternative k (ntt_plan_set_policy)."""
        check(_lib.lib().ntt_plan_set_policy(self._h, alternative), "ntt_plan_set_policy")

    def clone(self, device: int | None = None) -> "NTTPlan":
        """A copy of this plan on `device` (default: the same one): tables travel device-to-device (ntt_plan_clone)."""
        device = self.device if device is None else device
        new = object.__new__(NTTPlan)
        new.logn, new.n, new.p, new.word_bytes, new.device = self.logn, self.n, self.p, self.word_bytes, device
        new._h = C.c_void_p()
        new.table = self.table
        check(_lib.lib().ntt_plan_clone(self._h, device, C.byref(new._h)), "ntt_plan_clone")
        return new

    @property
    def has_inverse(self) -> bool:
        return bool(_lib.lib().ntt_plan_info(self._h, 4))

    # ---- buffers ---------------------------------------------------------------
    def empty(self, batch: int) -> torch.Tensor:
        return torch.empty((batch, self.n), dtype=_torch_dtype(self.word_bytes),
                           device=torch.device("cuda", self.device))

    def _batch(self, *ts: torch.Tensor) -> int:
        for t in ts:
            if not t.is_cuda or t.device.index != self.device:
                raise ValueError("buffer is not on cuda:%d" % self.device)
            if not t.is_contiguous() or t.element_size() != self.word_bytes:
                raise ValueError("buffer must be contiguous with %d-byte words" % self.word_bytes)
            if t.numel() % self.n or t.numel() != ts[0].numel():
                raise ValueError("buffer sizes must be equal multiples of N")
        return ts[0].numel() // self.n

    @staticmethod
    def _stream(stream) -> int:
        if stream is None:
            stream = torch.cuda.current_stream()
        return stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream)

    def _out_like(self, inp: torch.Tensor, stream) -> torch.Tensor:
        """Output buffer for a launch on `stream`: allocated UNDER that stream, so that torch's caching allocator
        ties the block's reuse to the stream the kernels run on (allocating on the current stream and launching on
        another would let the allocator recycle the block while the transform is still writing it)."""
        if stream is None:
            return torch.empty_like(inp)
        if not hasattr(stream, "cuda_stream"):  # a raw hipStream_t handle
            stream = torch.cuda.ExternalStream(int(stream), device=inp.device)
        with torch.cuda.stream(stream):
            return torch.empty_like(inp)

    # ---- transforms ------------------------------------------------------------
    def forward(self, inp: torch.Tensor, out: torch.Tensor | None = None, layout: int = LAYOUT_NATURAL,
                stream=None) -> torch.Tensor:
        """The reference network (src/test.cpp:34-60) on every polynomial of `inp`."""
        out = self._out_like(inp, stream) if out is None else out
        b = self._batch(inp, out)
        check(_lib.lib().ntt_forward(self._h, inp.data_ptr(), out.data_ptr(), b, layout,
                                     self._stream(stream)), "ntt_forward")
        return out

    def forward_profile(self, inp: torch.Tensor, out: torch.Tensor | None = None,
                        layout: int = LAYOUT_NATURAL, stream=None) -> list[float]:
        """forward() with a hipEvent pair around every HBM pass; returns ms per pass (blocking)."""
        out = self._out_like(inp, stream) if out is None else out
        b = self._batch(inp, out)
        ms = (C.c_float * 8)()
        k = C.c_int(0)
        check(_lib.lib().ntt_forward_profile(self._h, inp.data_ptr(), out.data_ptr(), b, layout,
                                             self._stream(stream), ms, 8, C.byref(k)), "ntt_forward_profile")
        return [float(ms[i]) for i in range(k.value)]

    def inverse(self, inp: torch.Tensor, out: torch.Tensor | None = None, layout: int = LAYOUT_NATURAL,
                scale: bool = True, stream=None) -> torch.Tensor:
        out = self._out_like(inp, stream) if out is None else out
        b = self._batch(inp, out)
        check(_lib.lib().ntt_inverse(self._h, inp.data_ptr(), out.data_ptr(), b, layout, int(scale),
                                     self._stream(stream)), "ntt_inverse")
        return out

    def pointwise_mul(self, a: torch.Tensor, b: torch.Tensor, out: torch.Tensor | None = None,
                      scale: int = 1, stream=None) -> torch.Tensor:
        out = self._out_like(a, stream) if out is None else out
        n = self._batch(a, b, out)
        check(_lib.lib().ntt_pointwise_mul(self._h, a.data_ptr(), b.data_ptr(), out.data_ptr(), n, scale,
                                           self._stream(stream)), "ntt_pointwise_mul")
        return out

    def polymul_negacyclic(self, a: torch.Tensor, b: torch.Tensor, out: torch.Tensor | None = None,
                           stream=None) -> torch.Tensor:
        """c = a*b mod (x^N + 1, p); needs a kind-2 table.  a and b are overwritten."""
        out = a if out is None else out
        n = self._batch(a, b, out)
        check(_lib.lib().ntt_polymul_negacyclic(self._h, a.data_ptr(), b.data_ptr(), out.data_ptr(), n,
                                                self._stream(stream)), "ntt_polymul_negacyclic")
        return out

    def count_noncanonical(self, buf: torch.Tensor) -> int:
        """How many words of `buf` are >= p (the transforms require canonical residues)."""
        b = self._batch(buf)
        n = C.c_uint64(0)
        check(_lib.lib().ntt_count_noncanonical(self._h, buf.data_ptr(), b, C.byref(n)), "ntt_count_noncanonical")
        return int(n.value)

    def forward_stages(self, inp: torch.Tensor, stage: int, out: torch.Tensor | None = None,
                       stream=None) -> torch.Tensor:
        """Stages 0..stage only (the reference's test_stage hook, src/test.cpp:55-58, 67)."""
        out = self._out_like(inp, stream) if out is None else out
        b = self._batch(inp, out)
        check(_lib.lib().ntt_forward_stages(self._h, inp.data_ptr(), out.data_ptr(), b, stage,
                                            self._stream(stream)), "ntt_forward_stages")
        return out
