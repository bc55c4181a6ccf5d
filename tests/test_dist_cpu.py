"""world_size-2 rehearsal of the multi-GPU path on CPU (gloo): batch sharding and
the one collective of the design, the twiddle-table broadcast."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from ntt_aie_amd import dist as nd

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
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
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = 1 << 10
    want = int((np.arange(n, dtype=np.uint64) * 0x9E3779B97F4A7C15 + 12345).sum(dtype=np.uint64))
    want32 = int((np.arange(n, dtype=np.uint32) * 2654435761 + 7).sum(dtype=np.uint64))
    assert res[0][1] == (0, 2050) and res[1][1] == (2050, 4099)
    assert all(r[2] == want and r[3] == want32 for r in res)
