import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """`-m gpu` on a box without a GPU must FAIL, not pass vacuously by skipping everything (the driver reads the
    exit code).  A plain `pytest tests/` (no -m) on a CPU-only box skips the gpu-marked tests."""
    expr = (config.getoption("-m") or "").strip()
    if "not gpu" in expr:
        return
    if _gpu_present():
        return
    if "gpu" in expr:
        raise pytest.UsageError("-m gpu was requested but no GPU is visible (torch.cuda.is_available() is False): "
                                "the GPU parity tests cannot run here; use gpurun")
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import oracle_py

    oracle_py.build()
    return oracle_py


# ---- rendezvous ----------------------------------------------------------------------------------------------------------
# Every test that starts a torch.distributed job rendezvouses through a FILE in a fresh temporary directory
# (init_method="file://..."), never through a TCP port: a port that is probed, released and handed to the job can be taken by
# another process in between (round 4: EADDRINUSE in tests/test_gpu_dist.py), and a retry only hides the next occurrence.  A
# failed rendezvous is reported, not retried -- as the reference reports a failed launch (src/test.cpp:162-166).
def spawn_world(worker, world: int, extra_args: tuple = (), timeout: int = 240):
    """Start `world` processes of worker(rank, world, rdzv_file, queue, *extra_args) (torch.multiprocessing, spawn) and return
    their queue items sorted by rank.  A worker that fails puts ("FAILED", rank, text) instead of raising."""
    import tempfile

    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    with tempfile.TemporaryDirectory(prefix="ntt_rdzv_") as tmp:
        q = ctx.Queue()
        rdzv = os.path.join(tmp, "store")
        procs = [ctx.Process(target=worker, args=(r, world, rdzv, q) + tuple(extra_args)) for r in range(world)]
        for p in procs:
            p.start()
        try:
            items = [q.get(timeout=timeout) for _ in procs]
        finally:
            for p in procs:
                p.join(timeout=60)
                if p.is_alive():  # the exact children started above
                    p.terminate()
    failed = [i for i in items if isinstance(i, tuple) and i and i[0] == "FAILED"]
    assert not failed, "worker failed: %r" % (failed,)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return sorted(items, key=lambda i: i[0])


def init_gloo_or_report(rank: int, world: int, rdzv_file: str, q) -> bool:
    """In a spawned worker: gloo process group over a file rendezvous.  False (after reporting to the parent) when it failed."""
    import torch.distributed as dist

    try:
        from datetime import timedelta

        dist.init_process_group("gloo", init_method="file://" + rdzv_file, rank=rank, world_size=world,
                                timeout=timedelta(seconds=120))
        return True
    except Exception as e:
        q.put(("FAILED", rank, repr(e)))
        return False
