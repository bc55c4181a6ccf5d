"""One process, N devices: the C boundary's multi-GPU form as a host-side class.

The reference scatters its data to the tiles, broadcasts its ONE table to every tile and gathers the results below a single
host (src/aie2.py:83-115: object-fifo scatter / broadcast / join).  `MultiDevicePlan` is that shape across GPUs: one plan made
on the first device, copied to the others with ntt_plan_clone (tables travel device-to-device, hipMemcpyPeer over xGMI), one
stream per device, contiguous rows of the batch per device (dist.shard_rows: the same partition the process-per-GPU form uses).
There is no data-path exchange: a transform never spans devices.  `bench.py --gpus N --single-process` times this form.
"""
from __future__ import annotations

import numpy as np
import torch

from .dist import shard_rows
from .plan import LAYOUT_NATURAL, NTTPlan


class MultiDevicePlan:
    """`devices` may repeat an index (several clones on one GPU: how a one-GPU box rehearses the path)."""

    def __init__(self, logn: int, p: int, word_bytes: int | None = None, devices: list[int] | None = None):
        if devices is None:
            devices = list(range(torch.cuda.device_count()))
        if not devices:
            raise ValueError("no device")
        self.devices = list(devices)
        first = NTTPlan(logn, p, word_bytes, self.devices[0])
        self.plans = [first]
        self.logn, self.n, self.p, self.word_bytes = first.logn, first.n, first.p, first.word_bytes
        self.streams = []
        for d in self.devices:
            with torch.cuda.device(d):
                self.streams.append(torch.cuda.Stream(device=torch.device("cuda", d)))
        self._cloned = False

    # ---- the table: made once, broadcast by cloning -----------------------------------------------------------------
    def _clone_all(self) -> None:
        for pl in self.plans[1:]:
            pl.close()
        self.plans = [self.plans[0]] + [self.plans[0].clone(d) for d in self.devices[1:]]
        self._cloned = True

    def set_twiddles(self, T: np.ndarray) -> None:
        self.plans[0].set_twiddles(T)
        self._clone_all()

    def generate_twiddles(self, kind: int, g: int) -> None:
        self.plans[0].generate_twiddles(kind, g)
        self._clone_all()

    def make_table(self, kind: int, g: int) -> np.ndarray:
        return self.plans[0].make_table(kind, g)

    # ---- shards -------------------------------------------------------------------------------------------------------
    def rows(self, batch: int) -> list[tuple[int, int]]:
        """Row range [lo, hi) of the [batch][N] job held by each device (contiguous, sizes differ by at most one)."""
        return [shard_rows(batch, len(self.devices), i) for i in range(len(self.devices))]

    def scatter(self, host: np.ndarray) -> list[torch.Tensor]:
        """Host [batch][N] words -> one device buffer per device (asynchronous copies on each device's stream)."""
        host = np.ascontiguousarray(host)
        signed = host.view(np.int32 if host.dtype.itemsize == 4 else np.int64)
        out = []
        for (lo, hi), d, st in zip(self.rows(host.shape[0]), self.devices, self.streams):
            with torch.cuda.device(d), torch.cuda.stream(st):
                out.append(torch.from_numpy(signed[lo:hi]).to(torch.device("cuda", d), non_blocking=True))
        return out

    def gather(self, shards: list[torch.Tensor]) -> np.ndarray:
        self.synchronize()
        parts = [s.detach().cpu().contiguous().numpy() for s in shards]
        a = np.concatenate(parts) if parts else np.empty((0, self.n))
        return a.view(np.uint32 if a.dtype.itemsize == 4 else np.uint64)

    def synchronize(self) -> None:
        for st in self.streams:
            st.synchronize()

    # ---- transforms: one launch sequence per device, all devices busy at once -----------------------------------------
    def _each(self, fn, ins, outs):
        if not self._cloned:
            raise RuntimeError("set_twiddles / generate_twiddles first")
        if outs is None:
            outs = [None] * len(ins)
        if not (len(ins) == len(outs) == len(self.plans)):
            raise ValueError("one shard per device")
        res = []
        for pl, st, x, y in zip(self.plans, self.streams, ins, outs):
            res.append(fn(pl, x, y, st) if x.shape[0] else (y if y is not None else x))
        return res

    def forward(self, shards: list[torch.Tensor], outs: list[torch.Tensor] | None = None, layout: int = LAYOUT_NATURAL):
        return self._each(lambda pl, x, y, st: pl.forward(x, y, layout=layout, stream=st), shards, outs)

    def inverse(self, shards: list[torch.Tensor], outs: list[torch.Tensor] | None = None, layout: int = LAYOUT_NATURAL,
                scale: bool = True):
        return self._each(lambda pl, x, y, st: pl.inverse(x, y, layout=layout, scale=scale, stream=st), shards, outs)

    def close(self) -> None:
        for pl in self.plans:
            pl.close()
