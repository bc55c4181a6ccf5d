"""The RCCL leg of the multi-GPU path, executed on the one-GPU box: an `nccl` process group of world size 1
(init over RCCL, the twiddle-table broadcast on the DEVICE, barrier, destroy).  The data path has no collective
(batch rows are independent: DESIGN.md section 6), so the table broadcast IS the whole RCCL surface; with more
ranks only the number of receivers changes.  Reference analogue: the on-chip broadcast of the root table to
every tile, src/aie2.py:96-104."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _env() -> dict:
    """One rank, no MASTER_ADDR / MASTER_PORT: the process groups below rendezvous through a file (no TCP port to collide on;
    round 4's EADDRINUSE came from probing a port, releasing it and handing the number over)."""
    env = {k: v for k, v in os.environ.items() if k not in ("MASTER_ADDR", "MASTER_PORT")}
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


def test_bench_force_dist_nccl_world1(tmp_path):
    """bench.py with --force-dist: the nccl branch of bench.py and of dist.broadcast_table run end to end (file rendezvous)."""
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "2", "--warmup", "1", "--batch", "64",
         "--no-cpu-baseline", "--rdzv-file", str(tmp_path / "store")], capture_output=True, text=True, timeout=900, env=_env())
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["steps"] == 2
    assert d["config"]["table_broadcast"] == "nccl"  # the RCCL branch, not the single-process shortcut


def test_broadcast_table_under_nccl_group(tmp_path):
    """dist.broadcast_table + ShardedNTT under an nccl group: the table is staged on the device, broadcast by RCCL,
    and the plan built from the received words transforms exactly like one built from the host table."""
    script = tmp_path / "nccl_bcast.py"
    script.write_text(
        "import os, sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'oracle'))\n"
        "import numpy as np, torch, torch.distributed as dist\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', init_method='file://' + sys.argv[1], rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
        "assert dist.get_backend() == 'nccl'\n"
        "from ntt_aie_amd.dist import ShardedNTT, broadcast_table, shard_rows\n"
        "from ntt_aie_amd import NTTPlan, to_device, to_host\n"
        "import oracle_py as O\n"
        "p, logn = O.GOLDILOCKS, 16\n"
        "T = O.make_roots(1 << logn, p, 7, 8)\n"
        "got = broadcast_table(T, 1 << logn, 8, src=0, device=torch.device('cuda', 0))\n"
        "assert got.dtype == np.uint64 and np.array_equal(got, T)\n"
        "T4 = O.make_roots(4096, 12289, 11, 4)\n"
        "assert np.array_equal(broadcast_table(T4, 4096, 4, src=0, device=torch.device('cuda', 0)), T4)\n"
        "eng = ShardedNTT(logn, p, g=7, word_bytes=8, device=0)\n"
        "assert np.array_equal(eng.table, T) and eng.world == 1 and eng.rows(4096) == (0, 4096)\n"
        "a = (np.random.default_rng(3).integers(0, 2**63, size=(3, 1 << logn), dtype=np.uint64) %% np.uint64(p))\n"
        "f = eng.forward_local(to_device(a, 'cuda:0'))\n"
        "assert np.array_equal(to_host(f), O.ntt(a, T, p, nthreads=4))\n"
        "assert np.array_equal(to_host(eng.inverse_local(f)), a)\n"
        "dist.barrier(); torch.cuda.synchronize(); dist.destroy_process_group()\n"
        "print('NCCL_BCAST_OK')\n" % (ROOT, ROOT))
    out = subprocess.run([sys.executable, str(script), str(tmp_path / "store")], capture_output=True, text=True, timeout=900, env=_env())
    assert out.returncode == 0 and "NCCL_BCAST_OK" in out.stdout, (out.stdout + out.stderr)[-3000:]
