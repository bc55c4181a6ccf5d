"""Round-4 additions, on a real MI355X through the C-ABI:
  * the reference-held scalar vectors (tests/golden/literal_scalar_q3329.npz: outputs of the literal
    src/aie_core.cc:11-39 modadd / modsub / barrett_2k) through the HIP arithmetic -- pointwise product, the N = 2
    butterfly with T[1] = 1 (sum and difference) and with T[1] = b (product), 4-byte AND 8-byte words;
  * the exception wall and the alignment rule of the C-ABI (NTT_E_NOMEM under an address-space limit, NTT_E_ARG for a
    misaligned row view), ntt_plan_info 8 as the capacity of ntt_forward_profile;
  * one decomposition per product (ntt_plan_select's contract) where the two-operand launch crosses an alternative's threshold;
  * every rank of bench.py proves its shard and names its device; the one-process / N-device mode."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

GOLD = 0xFFFFFFFF00000001


@pytest.fixture(scope="module")
def eng():
    import torch

    import ntt_aie_amd as E

    assert torch.cuda.is_available()
    assert os.path.exists(E.LIB_PATH), "native library missing: the GPU tests must not pass without it"
    torch.cuda.set_device(0)
    return E


def _rand(batch, n, p, dt, seed):
    rng = np.random.default_rng(seed)
    return (rng.integers(0, 2**63, size=(batch, n), dtype=np.uint64) % np.uint64(p)).astype(dt)


# ---- item 2: rows a3 / a4 on reference-held vectors ---------------------------------------------------------------------
@pytest.mark.parametrize("wb", [4, 8])
def test_reference_scalar_vectors_through_the_hip_arithmetic(eng, wb):
    """4 096 (a, b) pairs and what the reference's own scalar modadd / modsub / barrett_2k (src/aie_core.cc:11-39, constants
    src/aie2.py:17-19) made of them.  Four HIP legs, every one compared with the FIXTURE (no oracle in between):
      (1) ntt_pointwise_mul, scale 1                          -> barrett
      (2) N = 2 transform, T[1] = 1: (a, b) -> (a + b, a - b) -> modadd, modsub   (the pass kernel's butterfly)
      (3) N = 2 transforms, T[1] = b: (a, 0) -> (a, a * b)    -> barrett          (the butterfly's product, one plan per b)
      (4) stage 0 of an N = 8192 network with T[4096 + i] = b_i on (a_i, 0) pairs (test_stage hook) -> barrett, all pairs at once
    4-byte words run FieldM32, 8-byte words the general-modulus FieldM64 (p = 3329 is not Goldilocks)."""
    f = np.load(os.path.join(GOLDEN, "literal_scalar_q3329.npz"))
    q = int(f["q"])
    dt = np.uint32 if wb == 4 else np.uint64
    a, b = f["ab"][:, 0].astype(dt), f["ab"][:, 1].astype(dt)
    want_add, want_sub, want_mul = f["modadd"].astype(dt), f["modsub"].astype(dt), f["barrett"].astype(dt)
    npairs = a.size
    assert npairs == 4096 and q == 3329
    # (1) word-by-word product: one "polynomial" of 4096 words
    pl = eng.NTTPlan(12, q, wb, 0)
    got = eng.to_host(pl.pointwise_mul(eng.to_device(a[None, :], "cuda:0"), eng.to_device(b[None, :], "cuda:0")))[0]
    assert np.array_equal(got, want_mul), "pointwise product vs barrett_2k"
    # (2) the butterfly's sum and difference: N = 2, T = [*, 1]
    p2 = eng.NTTPlan(1, q, wb, 0)
    p2.set_twiddles(np.array([1, 1], dtype=dt))
    out = eng.to_host(p2.forward(eng.to_device(np.stack([a, b], axis=1), "cuda:0")))
    assert np.array_equal(out[:, 0], want_add), "x + y vs modadd"
    assert np.array_equal(out[:, 1], want_sub), "x - y vs modsub"
    # ... and back: the inverse butterfly undoes it
    assert np.array_equal(eng.to_host(p2.inverse(eng.to_device(out, "cuda:0"))), np.stack([a, b], axis=1))
    # (3) the butterfly's product: (a, 0) -> (a, (a - 0) * T[1]) with T[1] = b, one plan per distinct b (every one for 4-byte
    # words; the first 256 for 8-byte words: the same kernel each time, only the scalar differs)
    distinct = np.unique(b)
    if wb == 8:
        distinct = distinct[:256]
    checked = 0
    for bv in distinct:
        if bv == 0:
            continue  # a zero twiddle is legal for the forward network but uninteresting: product 0
        rows = np.nonzero(b == bv)[0]
        pb = eng.NTTPlan(1, q, wb, 0)
        pb.set_twiddles(np.array([1, bv], dtype=dt))
        x = np.stack([a[rows], np.zeros(rows.size, dtype=dt)], axis=1)
        o = eng.to_host(pb.forward(eng.to_device(x, "cuda:0")))
        assert np.array_equal(o[:, 0], a[rows]) and np.array_equal(o[:, 1], want_mul[rows]), int(bv)
        checked += rows.size
        pb.close()
    assert checked > (3000 if wb == 4 else 300)
    # (4) all pairs at once through stage 0 of a larger network: pair i is (word 2i, word 2i+1), its twiddle T[N/2 + i]
    n = 2 * npairs
    T = np.ones(n, dtype=dt)
    T[n // 2:] = b
    p13 = eng.NTTPlan(13, q, wb, 0)
    # set_twiddles derives the inverse table too and refuses nothing for a zero entry (has_inverse just turns false)
    p13.set_twiddles(T)
    x = np.zeros((1, n), dtype=dt)
    x[0, 0::2] = a
    o = eng.to_host(p13.forward_stages(eng.to_device(x, "cuda:0"), 0))[0]
    assert np.array_equal(o[0::2], a) and np.array_equal(o[1::2], want_mul), "stage 0 products vs barrett_2k"
    # and sum / difference through the same hook: T = 1 everywhere
    p13.set_twiddles(np.ones(n, dtype=dt))
    x[0, 0::2], x[0, 1::2] = a, b
    o = eng.to_host(p13.forward_stages(eng.to_device(x, "cuda:0"), 0))[0]
    assert np.array_equal(o[0::2], want_add) and np.array_equal(o[1::2], want_sub)


# ---- item 5: the C-ABI's error contract ---------------------------------------------------------------------------------
def test_misaligned_row_view_is_refused_not_faulted(eng):
    """include/ntt_hip.h: data pointers must be 16-byte aligned; a row view of an N = 2 batch of 4-byte words is 8 bytes
    into the buffer.  Every transform entry point answers NTT_E_ARG before launching anything."""
    import torch

    from ntt_aie_amd import _lib

    L = _lib.lib()
    pl = eng.NTTPlan(1, 3329, 4, 0)
    pl.set_twiddles(np.array([1, 1], dtype=np.uint32))
    buf = torch.zeros((8, 2), dtype=torch.int32, device="cuda:0")
    out = torch.zeros_like(buf)
    base, row1 = buf.data_ptr(), buf[1:].data_ptr()
    assert base % 16 == 0 and row1 % 16 == 8
    s = torch.cuda.current_stream().cuda_stream
    assert L.ntt_forward(pl._h, row1, out.data_ptr(), 1, 0, s) == _lib.NTT_E_ARG
    assert L.ntt_forward(pl._h, base, out[1:].data_ptr(), 1, 0, s) == _lib.NTT_E_ARG
    assert L.ntt_inverse(pl._h, row1, out.data_ptr(), 1, 0, 1, s) == _lib.NTT_E_ARG
    assert L.ntt_pointwise_mul(pl._h, base, base, out[1:].data_ptr(), 1, 1, s) == _lib.NTT_E_ARG
    assert L.ntt_forward_stages(pl._h, row1, out.data_ptr(), 1, 0, s) == _lib.NTT_E_ARG
    # rows 0, 2, 4 ... ARE aligned; and a refused call left everything untouched
    assert L.ntt_forward(pl._h, buf[2:].data_ptr(), out[2:].data_ptr(), 6, 0, s) == 0
    torch.cuda.synchronize()
    assert int(out.abs().sum()) == 0
    assert b"misaligned" in L.ntt_error_string(_lib.NTT_E_ARG)


def test_host_allocation_failure_returns_nomem(tmp_path):
    """ntt_plan_set_twiddles stages N-word std::vectors on the host (4 x 1 GiB at logn 27).  With the address space capped
    (RLIMIT_AS, in a child process, after the plan and the GPU runtime are up) operator new throws std::bad_alloc: the guard
    returns NTT_E_NOMEM, the process lives, the plan still works at a size that fits."""
    code = r'''
import ctypes as C, os, resource, sys
sys.path.insert(0, %r)
import numpy as np, torch
import ntt_aie_amd as E
from ntt_aie_amd import _lib
L = _lib.lib()
torch.cuda.set_device(0)
big = E.NTTPlan(27, 0xFFFFFFFF00000001, 8, 0)          # device tables: 2 x 1 GiB + 0.5 GiB, allocated BEFORE the cap
small = E.NTTPlan(10, 0xFFFFFFFF00000001, 8, 0)
T = np.ones(1 << 27, dtype=np.uint64)                    # the caller's own table: 1 GiB, also before the cap
vm = int(next(l for l in open("/proc/self/status") if l.startswith("VmSize")).split()[1]) * 1024
resource.setrlimit(resource.RLIMIT_AS, (vm + (1 << 30), resource.RLIM_INFINITY))   # 1 GiB of headroom: the 4 GiB of staging cannot fit
rc = L.ntt_plan_set_twiddles(big._h, T.ctypes.data)
print("RC", rc, L.ntt_error_string(rc).decode())
rc2 = L.ntt_make_table(big._h, 0, 7, T.ctypes.data)
print("RC2", rc2)
small.set_twiddles(small.make_roots(7))
a = torch.arange(1024, dtype=torch.int64, device="cuda:0")[None, :].contiguous()
back = small.inverse(small.forward(a))
print("ALIVE", bool(torch.equal(back, a)))
''' % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "RC -9 out of host memory" in out.stdout, out.stdout
    assert "RC2 -9" in out.stdout and "ALIVE True" in out.stdout, out.stdout


def test_profile_capacity_is_plan_info_8(eng):
    """ntt_plan_info(plan, 8) = the largest pass count over all alternatives = a safe capacity for ntt_forward_profile whatever
    the batch (4-byte N = 2^22, p >= 2^30: the default decomposition has 2 passes, batch >= 3 runs 3).  Too small a capacity
    answers NTT_E_ARG and says how many are needed."""
    import torch

    from ntt_aie_amd import _lib

    L = _lib.lib()
    pl = eng.NTTPlan(22, 3221225473, 4, 0)
    pl.generate_twiddles(0, 5)
    assert int(L.ntt_plan_info(pl._h, 3)) == 2 and int(L.ntt_plan_info(pl._h, 8)) == 3
    x = torch.zeros((4, 1 << 22), dtype=torch.int32, device="cuda:0")
    ms, k = (C.c_float * 8)(), C.c_int(0)
    s = torch.cuda.current_stream().cuda_stream
    assert L.ntt_forward_profile(pl._h, x.data_ptr(), x.data_ptr(), 4, 0, s, ms, 2, C.byref(k)) == _lib.NTT_E_ARG and k.value == 3
    cap = int(L.ntt_plan_info(pl._h, 8))
    assert L.ntt_forward_profile(pl._h, x.data_ptr(), x.data_ptr(), 4, 0, s, ms, cap, C.byref(k)) == 0 and k.value == 3
    assert L.ntt_forward_profile(pl._h, x.data_ptr(), x.data_ptr(), 1, 0, s, ms, cap, C.byref(k)) == 0 and k.value == 2
    one = eng.NTTPlan(10, GOLD, 8, 0)
    assert int(L.ntt_plan_info(one._h, 8)) == 1


def test_product_selects_its_decomposition_once(eng, oracle):
    """ntt_plan_select(batch) is THE alternative a product of `batch` pairs runs -- also when its two operands are one
    [2*batch][N] buffer and the two-operand launch alone would cross an alternative's threshold (8-byte N = 2^13: 13 stages
    from batch 128; here batch 64 .. 127 with contiguous operands).  Same words either way; checked against the oracle with
    contiguous and separate operands, default policy and both pinned alternatives."""
    import torch

    p, logn = GOLD, 13
    n = 1 << logn
    pl = eng.NTTPlan(logn, p, 8, 0)
    T = pl.make_table(2, 7)
    pl.set_twiddles(T)
    from ntt_aie_amd import _lib

    L = _lib.lib()
    for batch in (64, 100, 127, 128):
        assert int(L.ntt_plan_select(pl._h, batch)) == (1 if batch >= 128 else 0)
        a, b = _rand(batch, n, p, np.uint64, batch), _rand(batch, n, p, np.uint64, batch + 1)
        rows = [0, 1, batch // 2, batch - 1]
        A, B = oracle.intt(a[rows], T, p, nthreads=4), oracle.intt(b[rows], T, p, nthreads=4)
        want = oracle.ntt(oracle.pointwise(A, B, p, n % p), T, p, nthreads=4)
        for policy in (-1, 0, 1):
            pl.set_policy(policy)
            both = eng.to_device(np.concatenate([a, b]), "cuda:0")  # d_b directly follows d_a: the one-launch-per-pass path
            c = pl.polymul_negacyclic(both[:batch], both[batch:])
            assert np.array_equal(eng.to_host(c)[rows], want), (batch, policy, "contiguous")
            c2 = pl.polymul_negacyclic(eng.to_device(a, "cuda:0"), eng.to_device(b, "cuda:0"))
            assert torch.equal(c2, c), (batch, policy, "separate")
        pl.set_policy(-1)


# ---- items 1, 6, 7: bench.py ----------------------------------------------------------------------------------------------
def _bench(args, env_extra=None, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return out, (json.loads(lines[-1]) if lines else None)


REHEARSAL = {"NTT_BENCH_ONE_DEVICE": "1", "NTT_BENCH_BACKEND": "gloo"}


def test_bench_every_rank_verifies_and_names_its_device():
    """`bench.py --gpus 2` (rehearsal: both ranks on cuda:0 over gloo): the line carries all_ranks_verified, the world size the
    process group saw, and one record per rank (rank, device, PCI bus id / uuid, its own ms per step, its two verdicts)."""
    out, d = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "256"], REHEARSAL)
    assert out.returncode == 0 and d is not None, out.stdout[-1000:] + out.stderr[-3000:]
    assert d["all_ranks_verified"] is True and d["world_size_seen"] == 2 and d["launch"] == "process-per-gpu"
    assert [r["rank"] for r in d["ranks"]] == [0, 1]
    for r in d["ranks"]:
        assert r["round_trip_identical"] is True and r["coefficient_sum_invariant"] is True and r["rows_sampled"] >= 8
        assert r["local_device"] == 0 and r["ms_per_step"] > 0 and (r["pci_bus_id"] or r["uuid"])
    assert d["distinct_devices"] == 1  # the rehearsal's truth: two ranks, ONE GPU -- an 8-GPU line must say 8
    assert d["verification"]["round_trip_identical_all"] and d["verification"]["coefficient_sum_invariant_all"]
    assert max(r["ms_per_step"] for r in d["ranks"]) <= d["ms_per_step"] * 1.001  # the line's time is the max over ranks


def test_bench_four_ranks_rehearsal_file_rendezvous():
    """`bench.py --gpus 4` starting its own four ranks over a FILE rendezvous (round 5: no port is probed; rehearsal: all on cuda:0
    over gloo, within the box's limit of 6 GPU processes): four shards of the job's rows, four identity records gathered in rank
    order, the aggregate over all four, and the waived distinctness rule said in the line."""
    out, d = _bench(["--gpus", "4", "--steps", "2", "--warmup", "1", "--batch", "64"], REHEARSAL)
    assert out.returncode == 0 and d is not None, out.stdout[-1000:] + out.stderr[-3000:]
    assert d["n_gpus"] == 4 and d["world_size_seen"] == 4 and [r["rank"] for r in d["ranks"]] == [0, 1, 2, 3]
    assert d["all_ranks_verified"] is True and d["distinct_devices"] == 1
    assert d["verification"]["distinctness_waived_one_device_rehearsal"] is True and d["verification"]["ranks_on_distinct_devices"] is None
    assert abs(d["value"] - 4 * 64 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-6
    assert "configs" not in d and "cpu_baseline" not in d  # N = 1 only


def test_bench_one_failing_rank_fails_the_job():
    """A rank whose check fails (test hook NTT_BENCH_INJECT_FAILURE=<rank>: that rank reports round_trip false) turns rank 0's
    line red and the launcher's exit code non-zero -- a timed but wrong rank can no longer hide behind rank 0."""
    out, d = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "128"], dict(REHEARSAL, NTT_BENCH_INJECT_FAILURE="1"))
    assert out.returncode != 0, out.stdout[-1000:]
    assert d is not None and d["all_ranks_verified"] is False
    assert [r["round_trip_identical"] for r in d["ranks"]] == [True, False]


def test_bench_single_process_n_devices():
    """`bench.py --gpus 2 --single-process` (rehearsal: both plan clones on device 0): one process, ntt_plan_clone, one stream
    and one shard per device, every device verified -- SURVEY 8(e)'s one-process form, timed."""
    out, d = _bench(["--gpus", "2", "--single-process", "--steps", "3", "--warmup", "1", "--batch", "256", "--no-cpu-baseline",
                     "--no-valu-floor"], {"NTT_BENCH_ONE_DEVICE": "1"})
    assert out.returncode == 0 and d is not None, out.stdout[-1000:] + out.stderr[-3000:]
    assert d["launch"] == "single-process" and d["n_gpus"] == 2 and d["all_ranks_verified"] is True and len(d["ranks"]) == 2
    assert abs(d["value"] - 2 * 256 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6
    assert d["value_by_device_events"] >= d["value"] * 0.999  # the wall clock contains the slowest device's events
    assert "hipMemcpyPeer" in d["config"]["table_broadcast"] and "cpu_baseline" not in d
    assert all(r["round_trip_identical"] and r["coefficient_sum_invariant"] for r in d["ranks"])
    # the node's shape, eight plan clones / streams / shards in one process (rehearsal: all on device 0): config 5's partition
    out, d = _bench(["--gpus", "8", "--single-process", "--steps", "2", "--warmup", "1", "--batch", "128", "--no-cpu-baseline",
                     "--no-valu-floor", "--no-inverse"], {"NTT_BENCH_ONE_DEVICE": "1"})
    assert out.returncode == 0 and d is not None, out.stdout[-1000:] + out.stderr[-3000:]
    assert d["n_gpus"] == 8 and len(d["ranks"]) == 8 and d["all_ranks_verified"] is True and [r["rank"] for r in d["ranks"]] == list(range(8))
    assert abs(d["value"] - 8 * 128 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-6
    # more devices than the box has, no rehearsal switch: refused before any work
    out, d = _bench(["--gpus", "64", "--single-process", "--steps", "1", "--warmup", "0"])
    assert out.returncode == 2 and d is None and "only" in out.stderr


def _walk(o, path=""):
    if isinstance(o, dict):
        for k, v in o.items():
            yield from _walk(v, path + "/" + k)
    elif isinstance(o, list):
        for i, v in enumerate(o):
            yield from _walk(v, "%s[%d]" % (path, i))
    else:
        yield path, o


def test_bench_json_contract_frac_step_and_gpu_input_sample():
    """The default line (N = 1): roofline.frac_step (bytes over the line's own ms_per_step, NOT clamped since round 5) stays within
    8 % of frac (bytes over the kernels' hipEvent time) -- a step cannot be shorter than its kernels, beyond the clock the chip held
    in each phase: the two are measured seconds apart under a power cap whose held clock moves by the run-to-run spread of the
    device copy (3 % and more); the CPU baseline ran on the GPU's own input rows, never beyond the batch, and says so; the single
    rank verified itself; BASELINE configs 2 and 4 are in the same line, each verified without the oracle (round 5).  Round 6: the
    line is numbers, not commentary -- below 6 KB, no explanatory keys, and no clock above the part's 2.4 GHz anywhere in it."""
    out, d = _bench(["--steps", "5", "--warmup", "2", "--no-valu-floor", "--no-inverse", "--cpu-threads", "4"])
    assert out.returncode == 0 and d is not None, out.stdout[-1000:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    assert len(line) < 6144, len(line)
    leaves = list(_walk(d))
    assert not [k for k, _ in leaves if k.rsplit("/", 1)[-1].split("[")[0] in ("what", "definition", "bound_note", "verdict")]
    clocks = [v for k, v in leaves if "clock" in k and "GHz" in k and isinstance(v, (int, float))]
    assert all(v <= 2.4 * 1.02 for v in clocks), clocks
    r = d["roofline"]
    assert "frac_step_uncapped" not in r and 0 < r["frac_step"] <= r["frac"] * 1.08 and r["frac"] <= r["frac_ceiling"]
    assert abs(r["frac_step"] / r["frac"] - 1) < 0.15  # step time and summed kernel time describe the same launches (5 steps: the clock the chip holds in each phase moves them a few per cent apart)
    assert abs(r["achieved_step"] - r["algorithmic_bytes_per_launch"] / (d["ms_per_step"] * 1e-3) / 1e9) / r["achieved_step"] < 1e-4  # (nested floats carry 5 digits)
    # the step against what two trips at this run's device-copy rate would take: a fraction of a floor, so below 1 up to clock noise
    assert 0.5 < r["frac_of_practical_hbm"] < 1.05 and abs(r["practical_hbm_floor_ms"] - 2 * r["device_copy"]["ms"]) < 1e-4 * r["practical_hbm_floor_ms"]
    assert r["bound"] in ("hbm", "valu", "power-cap", "unsaturated") and r["roofline_of_fields"] == "hbm" and r["bound_detail"]
    if r["valu"] is not None:  # counters of exactly these sources are committed: the vector ALU against its MEASURED throughput (tools/hw.py)
        v = r["valu"]
        assert "peak_cycles_per_wave_instr" not in v  # constants (tools/hw.py) are not repeated in the slim line; --explain has them
        assert 0.5 < v["frac_of_peak_at_held_clock"] < 1.0 and all(80 < c < 95 for c in v["peak_cycles_per_butterfly"])
        assert all(0 < f < 1.0 for f in v["frac_of_peak_at_held_clock_per_pass"])
        assert (r["bound"] == "valu") == (v["frac_of_peak_at_held_clock"] >= 0.95 and min(r["pass_stream_frac_of_device_copy"]) < 0.95)
    c = d["cpu_baseline"]
    assert c["sample_is_gpu_input"] is True and "GPU's own" in c["sample"] and c["kind"] == "port" and 16 <= c["sample_rows"] <= 4096
    assert c["rows_beyond_gpu_batch"] == 0
    assert d["all_ranks_verified"] is True and d["world_size_seen"] == 1 and d["ranks"][0]["rank"] == 0
    assert d["launch"] == "process-per-gpu"
    # configs 2 and 4, driver-observed: one entry each, verified, with the roofline keys of the headline
    assert [e["key"] for e in d["configs"]] == ["cfg2", "cfg4", "cfg5_shard"] and d["configs_all_verified"] is True
    c2, c4, c5 = d["configs"]
    assert c5["baseline_config"] == 5 and c5["key"] == "cfg5_shard" and "batch" not in c5 and c5["verified"] and c5["hbm_passes"] == 2 and c5["unit"] == "NTT/s"
    assert 0.5 * d["value"] < c5["value"] < 1.5 * d["value"]  # the shard runs at the headline's rate (twice the rows, twice the time)
    assert c2["baseline_config"] == 2 and c2["verified"] and c2["verification"]["round_trip_identical"] and c2["verification"]["coefficient_sum_invariant"]
    assert c2["unit"] == "NTT/s" and 0 < c2["ms"] < 1.0 and abs(c2["value"] - 1024 / (c2["ms"] * 1e-3)) / c2["value"] < 1e-4
    assert c4["baseline_config"] == 4 and c4["verified"] and c4["verification"]["evaluation_at_root_of_xN_plus_1"] and c4["verification"]["transform_domain_identity_whole_batch"]
    assert c4["unit"] == "products/s" and 1.0 < c4["ms"] < 100.0
    for e in (c2, c4, c5):
        rr = e["roofline"]
        assert 0 < rr["frac"] <= rr["frac_ceiling"] <= 1.0 and rr["peak"] == 8000.0 and rr["unit"] == "GB/s" and rr["bound"]
        assert abs(rr["achieved"] - rr["algorithmic_bytes_per_op"] / (e["ms"] * 1e-3) / 1e9) / rr["achieved"] < 1e-4


# ---- the wide radix-8 variant of the 4-byte single-pass sizes (plan.h: PassDesc::variant 1) ---------------------------------
def _check_all_legs(eng, oracle, pl, T, p, dt, batches, seed):
    n = pl.n
    for batch in batches:
        a = _rand(batch, n, p, dt, seed + batch)
        want = oracle.ntt(a, T, p, nthreads=8)
        d = eng.to_device(a, "cuda:0")
        f = pl.forward(d)
        assert np.array_equal(eng.to_host(f), want), ("forward", batch)
        blk = pl.forward(d, layout=eng.LAYOUT_AIE_BLOCK16)
        assert np.array_equal(eng.to_host(blk), oracle.block16(want)), ("block16", batch)
        assert np.array_equal(eng.to_host(pl.inverse(blk, layout=eng.LAYOUT_AIE_BLOCK16)), a), ("inverse of block16", batch)
        assert np.array_equal(eng.to_host(pl.inverse(f)), a), ("inverse", batch)
        u = eng.to_host(pl.inverse(f, scale=False)).astype(object)
        assert np.array_equal((u % p).astype(dt), ((a.astype(object) * n) % p).astype(dt)) and int(u.max()) < p, ("unscaled inverse", batch)
        g = d.clone()
        pl.forward(g, g)
        pl.inverse(g, g)
        assert np.array_equal(eng.to_host(g), a), ("in place round trip", batch)


@pytest.mark.parametrize("wb,p,g", [(4, 3221225473, 5), (4, 2013265921, 31), (4, 998244353, 3), (4, 3329, 3), (8, GOLD, 7),
                                     (8, 0x3FFFFFEE00000001, 3)])
def test_wide_variant_of_single_pass_sizes(eng, oracle, wb, p, g):
    """N = 2^10 .. 2^12, both word widths: alternative 0 runs the unit on 512 threads x 8 words (variant 1) below a batch
    threshold, alternative 1 the default radix-16 kernel from it on.  Both pinned and through the batch rule: forward, both layouts
    (AIE_BLOCK16 straight from the radix-8 kernel: one address bit moves between the register index and the thread), inverse (8-byte: N^-1 folded into stage 0 of the wide kernel too), unscaled
    inverse, in place, ragged batches, every 4-byte modulus class, Goldilocks and the general 64-bit modulus; the negacyclic
    product under both (its fused kernel is its own)."""
    from ntt_aie_amd import _lib

    L = _lib.lib()
    dt = np.uint32 if wb == 4 else np.uint64
    for logn in (10, 11, 12):
        n = 1 << logn
        T = oracle.make_roots(n, p, g, wb)
        pl = eng.NTTPlan(logn, p, wb, 0)
        pl.set_twiddles(T)
        alts = pl.alternatives
        assert [a[0] for a in alts] == [[logn], [logn]] and alts[0][1] == 0 and alts[1][1] >= 256
        assert pl.alternative_variants == [[1], [0]]
        thr = alts[1][1]
        assert int(L.ntt_plan_select(pl._h, thr - 1)) == 0 and int(L.ntt_plan_select(pl._h, thr)) == 1
        for k in (0, 1):
            pl.set_policy(k)
            _check_all_legs(eng, oracle, pl, T, p, dt, (1, 3, 37, 300), seed=logn * 10 + k)
        pl.set_policy(-1)
        for batch in (thr - 1, thr):
            a = _rand(batch, n, p, dt, batch)
            f = pl.forward(eng.to_device(a, "cuda:0"))
            rows = [0, 1, batch // 2, batch - 1]
            assert np.array_equal(eng.to_host(f)[rows], oracle.ntt(a[rows], T, p, nthreads=4)), batch
            assert np.array_equal(eng.to_host(pl.inverse(f)), a), batch
    if (p - 1) % (1 << 13) == 0:  # a negacyclic table exists: the product under both alternatives
        logn, n = 12, 4096
        pl = eng.NTTPlan(logn, p, wb, 0)
        T2 = pl.make_table(2, g)
        pl.set_twiddles(T2)
        a, b = _rand(5, n, p, dt, 1), _rand(5, n, p, dt, 2)
        A, B = oracle.intt(a, T2, p), oracle.intt(b, T2, p)
        want = oracle.ntt(oracle.pointwise(A, B, p, n % p), T2, p)
        for k in (0, 1):
            pl.set_policy(k)
            c = pl.polymul_negacyclic(eng.to_device(a, "cuda:0"), eng.to_device(b, "cuda:0"))
            assert np.array_equal(eng.to_host(c), want), k


def test_multi_device_plan_scatter_transform_gather(eng, oracle):
    """ntt_aie_amd.multi.MultiDevicePlan: one process, one plan clone + one stream per device, contiguous rows per device
    (three clones on device 0 here).  Scatter a ragged [37][N] job, transform every shard, gather: the oracle's words; the
    inverse gives the input back; empty shards (more devices than rows) are no-ops."""
    from ntt_aie_amd import MultiDevicePlan

    p, logn = GOLD, 13
    n = 1 << logn
    md = MultiDevicePlan(logn, p, 8, devices=[0, 0, 0])
    with pytest.raises(RuntimeError):
        md.forward([])
    T = md.make_table(0, 7)
    md.set_twiddles(T)
    assert md.rows(37) == [(0, 13), (13, 25), (25, 37)] and len(md.plans) == 3 and len({id(s) for s in md.streams}) == 3
    a = _rand(37, n, p, np.uint64, 5)
    shards = md.scatter(a)
    assert [s.shape[0] for s in shards] == [13, 12, 12]
    f = md.forward(shards)
    assert np.array_equal(md.gather(f), oracle.ntt(a, T, p, nthreads=8))
    assert np.array_equal(md.gather(md.inverse(f)), a)
    blk = md.forward(shards, layout=eng.LAYOUT_AIE_BLOCK16)
    assert np.array_equal(md.gather(blk), oracle.block16(oracle.ntt(a, T, p, nthreads=8)))
    two = md.scatter(a[:2])  # rows (0,1), (1,2), (2,2): the last device holds nothing
    assert [s.shape[0] for s in two] == [1, 1, 0]
    assert np.array_equal(md.gather(md.forward(two)), oracle.ntt(a[:2], T, p))
    md.generate_twiddles(0, 7)  # re-made on the first device and cloned again
    assert np.array_equal(md.gather(md.forward(shards)), oracle.ntt(a, T, p, nthreads=8))
    md.close()
