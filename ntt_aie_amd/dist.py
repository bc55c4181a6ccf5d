"""Multi-GPU batch sharding: one process per GPU, no data-path collective.

Polynomials of a batch are independent (the reference never splits one
transform across devices either: its 16 tiles share ONE vector,
src/aie2.py:21-26), so rank r owns the contiguous rows
[r*B/W, (r+1)*B/W) of the [B][N] buffer and transforms them locally.  The only
exchange is the twiddle table ("root" buffer, src/test.cpp:137-143): rank 0
makes it and broadcasts N words once at plan creation -- over RCCL/xGMI with the
`nccl` backend, over `gloo` in the CPU rehearsal.  It plays the role of the
reference's on-chip broadcast of the table to every tile (src/aie2.py:96-104).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def shard_rows(batch: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous row range [lo, hi) of rank `rank`; the first batch % world ranks get one extra row."""
    base, extra = divmod(batch, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def broadcast_table(table: np.ndarray | None, n: int, word_bytes: int, src: int = 0,
                    device: torch.device | None = None) -> np.ndarray:
    """Every rank returns the table held by rank `src` (N words).

    With the nccl (= RCCL) backend the payload travels GPU-to-GPU over xGMI;
    with gloo it travels through host memory.
    """
    np_dt = np.uint32 if word_bytes == 4 else np.uint64
    t_dt = torch.int32 if word_bytes == 4 else torch.int64
    if not (dist.is_available() and dist.is_initialized()):
        if table is None:
            raise ValueError("single process: the table must be supplied")
        return np.ascontiguousarray(table, dtype=np_dt)
    on_gpu = dist.get_backend() == "nccl"
    dev = device if on_gpu else torch.device("cpu")
    if dist.get_rank() == src:
        if table is None:
            raise ValueError("source rank must supply the table")
        host = np.ascontiguousarray(table, dtype=np_dt)
        buf = torch.from_numpy(host.view(np.int32 if word_bytes == 4 else np.int64).copy()).to(dev)
    else:
        buf = torch.empty(n, dtype=t_dt, device=dev)
    dist.broadcast(buf, src=src)
    return buf.cpu().numpy().view(np_dt).copy()


class ShardedNTT:
    """Plan + local shard bookkeeping for one rank."""

    def __init__(self, logn: int, p: int, g: int | None = None, table: np.ndarray | None = None,
                 word_bytes: int | None = None, table_kind: int = 0, device: int | None = None):
        from .plan import NTTPlan

        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.plan = NTTPlan(logn, p, word_bytes, device)
        if self.rank == 0 and table is None:
            if g is None:
                raise ValueError("rank 0 needs a generator g or a table")
            table = self.plan.make_table(table_kind, g)
        dev = torch.device("cuda", self.plan.device)
        self.table = broadcast_table(table if self.rank == 0 else None, self.plan.n, self.plan.word_bytes,
                                     src=0, device=dev)
        self.plan.set_twiddles(self.table)

    def rows(self, batch: int) -> tuple[int, int]:
        return shard_rows(batch, self.world, self.rank)

    def forward_local(self, local_in: torch.Tensor, local_out: torch.Tensor | None = None, **kw):
        return self.plan.forward(local_in, local_out, **kw)

    def inverse_local(self, local_in: torch.Tensor, local_out: torch.Tensor | None = None, **kw):
        return self.plan.inverse(local_in, local_out, **kw)
