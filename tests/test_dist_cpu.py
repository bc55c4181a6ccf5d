"""world_size-2 and world_size-8 rehearsals of the multi-GPU path on CPU (gloo, file rendezvous): batch sharding and
the one collective of the design, the twiddle-table broadcast."""
import os
import sys

import numpy as np

from conftest import ROOT, spawn_world


def _worker(rank, world, rdzv, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    from conftest import init_gloo_or_report
    from ntt_aie_amd import dist as nd

    if not init_gloo_or_report(rank, world, rdzv, q):
        return
    n = 1 << 10
    table = (np.arange(n, dtype=np.uint64) * 0x9E3779B97F4A7C15 + 12345) if rank == 0 else None
    got = nd.broadcast_table(table, n, 8, src=0)
    t32 = (np.arange(n, dtype=np.uint32) * 2654435761 + 7) if rank == 0 else None
    got32 = nd.broadcast_table(t32, n, 4, src=0)
    # config 5's shape: 65536 rows over the job (8 x 8192 on the node), and a ragged batch
    q.put((rank, nd.shard_rows(4099, world, rank), int(got.sum(dtype=np.uint64)), int(got32.sum(dtype=np.uint64)),
           nd.shard_rows(65536, world, rank)))
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


def test_gloo_world8_broadcast_and_config5_shards():
    """The 8-rank shapes of the node the driver scales to (src/aie2.py:83-115 scatter / broadcast / gather below one
    host): the table broadcast with SEVEN receivers, and BASELINE config 5's partition 65536 = 8 x 8192 contiguous rows."""
    res = spawn_world(_worker, 8)
    n = 1 << 10
    want = int((np.arange(n, dtype=np.uint64) * 0x9E3779B97F4A7C15 + 12345).sum(dtype=np.uint64))
    want32 = int((np.arange(n, dtype=np.uint32) * 2654435761 + 7).sum(dtype=np.uint64))
    assert [r[0] for r in res] == list(range(8))
    assert all(r[2] == want and r[3] == want32 for r in res)  # every receiver holds rank 0's words
    assert [r[4] for r in res] == [(8192 * k, 8192 * (k + 1)) for k in range(8)]
    edges = [r[1] for r in res]  # ragged: 4099 = 3 x 513 + 5 x 512
    assert edges[0][0] == 0 and edges[-1][1] == 4099 and all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
    assert sorted(b - a for a, b in edges) == [512] * 5 + [513] * 3
