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


# ---- rendezvous ports ---------------------------------------------------------------------------------------------------
# A test that starts a torch.distributed job picks a free TCP port, closes the probe socket and hands the number to the job: between
# the two another process (a rank of the previous test that is still shutting down) can take it, and the job dies with EADDRINUSE.
# Every such test goes through run_with_fresh_port(): the job is started again on a new port when -- and only when -- that is why
# it failed.
def free_port() -> int:
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def port_collision(text: str) -> bool:
    return "EADDRINUSE" in text or "address already in use" in text.lower()


def run_with_fresh_port(start, attempts: int = 4):
    """start(port) -> subprocess.CompletedProcess (stdout / stderr captured as text).  Returns the first result that is not a port
    collision (or the last one)."""
    out = None
    for _ in range(attempts):
        out = start(free_port())
        if out.returncode == 0 or not port_collision((out.stdout or "") + (out.stderr or "")):
            return out
    return out


def spawn_world(worker, world: int, extra_args: tuple = (), attempts: int = 4, timeout: int = 180):
    """Start `world` processes of worker(rank, world, port, queue, *extra_args) (torch.multiprocessing, spawn) and return their
    queue items sorted.  A worker reports a failed rendezvous by putting ("RENDEZVOUS_FAILED", rank, text) instead of raising;
    when that text is a port collision the whole world is started again on a fresh port."""
    import torch.multiprocessing as mp

    last = None
    for _ in range(attempts):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = free_port()
        procs = [ctx.Process(target=worker, args=(r, world, port, q) + tuple(extra_args)) for r in range(world)]
        for p in procs:
            p.start()
        items = [q.get(timeout=timeout) for _ in procs]
        for p in procs:
            p.join(timeout=60)
        failed = [i for i in items if isinstance(i, tuple) and i and i[0] == "RENDEZVOUS_FAILED"]
        if not failed:
            assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
            return sorted(items, key=lambda i: i[0])
        last = failed
        if not any(port_collision(f[2]) for f in failed):
            break
    raise AssertionError("rendezvous failed: %r" % (last,))


def init_gloo_or_report(rank: int, world: int, port: int, q) -> bool:
    """In a spawned worker: gloo process group on 127.0.0.1:port.  False (after reporting to the parent) when the rendezvous failed."""
    import os

    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    try:
        from datetime import timedelta

        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=60))
        return True
    except Exception as e:  # the parent decides whether to start the world again
        q.put(("RENDEZVOUS_FAILED", rank, repr(e)))
        return False
