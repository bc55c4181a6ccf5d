"""world_size-2 rehearsal of the multi-GPU path on CPU (gloo): batch sharding and
the one collective of the design, the twiddle-table broadcast."""
import os
import sys

import numpy as np

from conftest import ROOT, spawn_world


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    from conftest import init_gloo_or_report
    from ntt_aie_amd import dist as nd

    if not init_gloo_or_report(rank, world, port, q):
        return
    n = 1 << 10
    table = (np.arange(n, dtype=np.uint64) * 0x9E3779B97F4A7C15 + 12345) if rank == 0 else None
    got = nd.broadcast_table(table, n, 8, src=0)
    t32 = (np.arange(n, dtype=np.uint32) * 2654435761 + 7) if rank == 0 else None
    got32 = nd.broadcast_table(t32, n, 4, src=0)
    q.put((rank, nd.shard_rows(4099, world, rank), int(got.sum(dtype=np.uint64)), int(got32.sum(dtype=np.uint64))))
    dist.destroy_process_group()


def test_shard_rows_partition():
    from ntt_aie_amd.dist import shard_rows

    for batch in (0, 1, 7, 8, 4096, 4099, 65536):
        for world in (1, 2, 3, 8):
            edges = [shard_rows(batch, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == batch
            for (a, b), (c, d) in zip(edges, edges[1:]):
                assert b == c and a <= b
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1


def test_gloo_world2_broadcast_and_shards():
    res = spawn_world(_worker, 2)
    n = 1 << 10
    want = int((np.arange(n, dtype=np.uint64) * 0x9E3779B97F4A7C15 + 12345).sum(dtype=np.uint64))
    want32 = int((np.arange(n, dtype=np.uint32) * 2654435761 + 7).sum(dtype=np.uint64))
    assert res[0][1] == (0, 2050) and res[1][1] == (2050, 4099)
    assert all(r[2] == want and r[3] == want32 for r in res)
