"""Round-3 additions, on a real MI355X through the C-ABI, bit-exact against the oracle:
  * every launch-time plan alternative (plan.h: plan_alternatives) -- forward, inverse, unscaled inverse, product, ragged
    batches -- pinned with ntt_plan_set_policy AND reached through the batch rule;
  * the 9-stage column pass (N = 2^22 in two passes) and the 14-stage 4-byte pass;
  * the scaled inverse with N^-1 folded into stage 0 (every Goldilocks CONTIG kernel shape; host-made and device-made tables);
  * ntt_plan_clone and the C++ multi-device host (tests/cxx/multi_device_host.cpp);
  * bench.py starting its own ranks (`python bench.py --gpus 2`, rehearsal mode on the one-GPU box)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

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


def _alts(pl):
    """[(stages per pass, min_batch)] of a plan, through ntt_plan_info."""
    from ntt_aie_amd import _lib

    L = _lib.lib()
    out = []
    for a in range(int(L.ntt_plan_info(pl._h, 6))):
        k = int(L.ntt_plan_info(pl._h, 256 + 16 * a))
        out.append(([int(L.ntt_plan_info(pl._h, 256 + 16 * a + 1 + i)) for i in range(k)], int(L.ntt_plan_info(pl._h, 256 + 16 * a + 15))))
    return out


def _check_all_legs(eng, oracle, pl, T, p, dt, batches, seed, product_table=None):
    """forward / block order / inverse / unscaled inverse / in place, on ragged batches, against the oracle."""
    n = pl.n
    for batch in batches:
        a = _rand(batch, n, p, dt, seed + batch)
        want = oracle.ntt(a, T, p, nthreads=8)
        d = eng.to_device(a, "cuda:0")
        f = pl.forward(d)
        assert np.array_equal(eng.to_host(f), want), ("forward", batch)
        if n >= 16:
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


CASES = [  # (word bytes, p, g, logn, expected alternatives as stage lists)
    (8, GOLD, 7, 13, [[7, 6], [13]]),
    (4, 998244353, 3, 14, [[8, 6], [14]]),
    (4, 3329, 3, 14, [[8, 6], [14]]),
]


@pytest.mark.parametrize("wb,p,g,logn,expect", CASES)
def test_every_plan_alternative(eng, oracle, wb, p, g, logn, expect):
    dt = np.uint32 if wb == 4 else np.uint64
    n = 1 << logn
    T = oracle.make_roots(n, p, g, wb)
    pl = eng.NTTPlan(logn, p, wb, 0)
    pl.set_twiddles(T)
    alts = _alts(pl)
    assert [a[0] for a in alts] == expect and alts[0][1] == 0 and alts[1][1] > 1
    for k in range(len(alts)):
        pl.set_policy(k)
        assert [m for _, _, m in pl.passes_for(1)] == expect[k] == [m for _, _, m in pl.passes_for(1 << 20)]
        _check_all_legs(eng, oracle, pl, T, p, dt, (1, 3, 37), seed=logn * 10 + k)
    # ... and through the batch rule: below the threshold the default, at it the long pass
    pl.set_policy(-1)
    thr = alts[1][1]
    assert [m for _, _, m in pl.passes_for(thr - 1)] == expect[0] and [m for _, _, m in pl.passes_for(thr)] == expect[1]
    for batch in (thr - 1, thr, thr + 5):
        a = _rand(batch, n, p, dt, batch)
        f = pl.forward(eng.to_device(a, "cuda:0"))
        rows = [0, 1, batch // 2, batch - 1]
        assert np.array_equal(eng.to_host(f)[rows], oracle.ntt(a[rows], T, p, nthreads=4)), batch
        assert np.array_equal(eng.to_host(pl.inverse(f)), a), batch
        assert len(pl.forward_profile(eng.to_device(a, "cuda:0"))) == len(pl.passes_for(batch))


@pytest.mark.parametrize("wb,p,g,logn", [(8, GOLD, 7, 13), (4, 998244353, 3, 14)])
def test_product_under_every_alternative(eng, oracle, wb, p, g, logn):
    """ntt_polymul_negacyclic with a pinned alternative: the fused middle pass where its unit exists (2^13 of 4-byte words, the
    7-stage Goldilocks unit), the separate passes otherwise (13-stage Goldilocks unit, 14-stage 4-byte unit) -- same words."""
    dt = np.uint32 if wb == 4 else np.uint64
    n = 1 << logn
    pl = eng.NTTPlan(logn, p, wb, 0)
    T = pl.make_table(2, g)
    pl.set_twiddles(T)
    a, b = _rand(5, n, p, dt, 1), _rand(5, n, p, dt, 2)
    # the oracle pipeline: c = Fwd(Inv(a) . Inv(b) . N) with the kind-2 table (oracle.intt scales by N^-1)
    A, B = oracle.intt(a, T, p), oracle.intt(b, T, p)
    want = oracle.ntt(oracle.pointwise(A, B, p, n % p), T, p)
    for k in range(len(_alts(pl))):
        pl.set_policy(k)
        c = pl.polymul_negacyclic(eng.to_device(a, "cuda:0"), eng.to_device(b, "cuda:0"))
        assert np.array_equal(eng.to_host(c), want), k
        # (x^(n-1)) * x = x^n = -1
        ea, eb = np.zeros((1, n), dtype=dt), np.zeros((1, n), dtype=dt)
        ea[0, n - 1], eb[0, 1] = 1, 1
        c = eng.to_host(pl.polymul_negacyclic(eng.to_device(ea, "cuda:0"), eng.to_device(eb, "cuda:0")))
        e = np.zeros((1, n), dtype=dt)
        e[0, 0] = p - 1
        assert np.array_equal(c, e), k


@pytest.mark.parametrize("wb,p,g", [(8, GOLD, 7), (4, 3221225473, 5), (4, 998244353, 3),
                                     (8, 0x3FFFFFEE00000001, 3), (8, 0xFFFFFFFC00000001, 10)])
def test_two_pass_2p22_nine_stage_column_pass(eng, oracle, wb, p, g):
    """N = 2^22 = 13 + 9: the 512-row column tile (512 / 1024 threads), forward / inverse / block order, ragged batch.
    Round 4: the general 64-bit modulus too -- ColCfg<9, FieldM64> in 512 threads under the 128-VGPR cap with
    the m64 streams' pinned scratch, and the forward ContigCfg13<FieldM64>, had only host-model coverage."""
    dt = np.uint32 if wb == 4 else np.uint64
    logn, n = 22, 1 << 22
    T = oracle.make_roots(n, p, g, wb)
    pl = eng.NTTPlan(logn, p, wb, 0)
    pl.set_twiddles(T)
    assert [m for _, _, m in pl.passes] == [13, 9] and pl.hbm_passes == 2
    _check_all_legs(eng, oracle, pl, T, p, dt, (1, 3), seed=22)


@pytest.mark.parametrize("ov", ["8,9", "5,9", "9,9", "12,9"])
def test_nine_stage_column_pass_other_offsets(eng, oracle, ov):
    """The 9-stage column kernel at other first stages (experiment build: NTT_PLAN_SPLIT), both word sizes."""
    exp = os.path.join(ROOT, "ntt_aie_amd", "libntt_hip_exp.so")
    if not os.path.exists(exp):
        pytest.skip("experiment library not built")
    code = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "oracle")); sys.path.insert(0, os.path.join(%r, "tools"))
import numpy as np
import _explib
_explib.select()
os.environ["NTT_PLAN_SPLIT"] = %r
import torch, ntt_aie_amd as E, oracle_py as O
logn = sum(int(x) for x in %r.split(","))
for wb, p, g in ((8, 0xFFFFFFFF00000001, 7), (4, 3221225473, 5), (4, 998244353, 3)):
    dt = np.uint32 if wb == 4 else np.uint64
    n = 1 << logn
    T = O.make_roots(n, p, g, wb)
    pl = E.NTTPlan(logn, p, wb, 0); pl.set_twiddles(T)
    assert [m for _, _, m in pl.passes] == [int(x) for x in %r.split(",")], pl.passes
    rng = np.random.default_rng(logn)
    a = (rng.integers(0, 2**63, size=(3, n), dtype=np.uint64) %% np.uint64(p)).astype(dt)
    f = pl.forward(E.to_device(a, "cuda:0"))
    assert np.array_equal(E.to_host(f), O.ntt(a, T, p, nthreads=8)), (wb, "forward")
    assert np.array_equal(E.to_host(pl.inverse(f)), a), (wb, "inverse")
    blk = pl.forward(E.to_device(a, "cuda:0"), layout=E.LAYOUT_AIE_BLOCK16)
    assert np.array_equal(E.to_host(blk), O.block16(O.ntt(a, T, p, nthreads=8))), (wb, "block16")
print("OK")
''' % (ROOT, ROOT, ROOT, ov, ov, ov)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@pytest.mark.parametrize("p,g", [(GOLD, 7), (0x3FFFFFEE00000001, 3)])
def test_scaled_inverse_fold_every_contig_shape(eng, oracle, p, g):
    """N^-1 folded into stage 0 (pass.h: fold_scale): every 8-byte CONTIG kernel shape as the LAST inverse pass -- radix-16
    single-pass sizes 2^1..2^12, radix-8 first passes of 7..12 stages (N = 2^13 .. 2^20), the 13-stage pass (2^21, 2^13 alt) --
    with host-made tables (ntt_plan_set_twiddles) and device-made ones (ntt_plan_generate_twiddles); edge residues included.
    Goldilocks and the general 64-bit modulus (both fields fold)."""
    dt = np.uint64
    for logn in list(range(1, 13)) + [13, 14, 15, 16, 17, 18, 19, 20, 21]:
        n = 1 << logn
        batch = 5 if logn <= 16 else 2
        T = oracle.make_roots(n, p, g, 8)
        a = _rand(batch, n, p, dt, logn)
        a[0, : min(n, 8)] = np.array([0, 1, p - 1, p - 2, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFF00000000 % p, 2][: min(n, 8)], dtype=dt)
        f = oracle.ntt(a, T, p, nthreads=8)
        for made in ("host", "device"):
            pl = eng.NTTPlan(logn, p, 8, 0)
            if made == "host":
                pl.set_twiddles(T)
            else:
                pl.generate_twiddles(0, g)
            for k in range(len(_alts(pl))):
                pl.set_policy(k)
                back = eng.to_host(pl.inverse(eng.to_device(f, "cuda:0")))
                assert np.array_equal(back, a), (logn, made, k)
    # an arbitrary (non-power) invertible table: the scaled table must follow T^-1, not a generator
    logn, n = 10, 1024
    rng = np.random.default_rng(5)
    T = (rng.integers(1, 2**63, size=n, dtype=np.uint64) % np.uint64(p - 1) + np.uint64(1)).astype(dt)
    pl = eng.NTTPlan(logn, p, 8, 0)
    pl.set_twiddles(T)
    a = _rand(3, n, p, dt, 9)
    assert np.array_equal(eng.to_host(pl.inverse(pl.forward(eng.to_device(a, "cuda:0")))), a)


def test_plan_clone(eng, oracle):
    """ntt_plan_clone: tables copied device-to-device (same device here; hipMemcpyPeer across devices), policy and
    alternatives travel, the source may be destroyed afterwards; a clone of a table-less plan is table-less."""
    from ntt_aie_amd import _lib

    L = _lib.lib()
    for wb, p, g, logn in ((8, GOLD, 7, 13), (4, 998244353, 3, 12)):
        dt = np.uint32 if wb == 4 else np.uint64
        n = 1 << logn
        src = eng.NTTPlan(logn, p, wb, 0)
        empty = src.clone(0)
        with pytest.raises(eng.NTTError):
            empty.forward(eng.to_device(_rand(1, n, p, dt, 0), "cuda:0"))
        src.generate_twiddles(0, g)
        if wb == 8:
            src.set_policy(1)
        cl = src.clone(0)
        assert cl.device == 0 and np.array_equal(cl.get_twiddles(), src.get_twiddles())
        assert np.array_equal(cl.get_twiddles(inverse=True), src.get_twiddles(inverse=True))
        assert int(L.ntt_plan_info(cl._h, 7)) == (1 if wb == 8 else -1)
        T = src.get_twiddles()
        src.close()
        a = _rand(4, n, p, dt, logn)
        f = cl.forward(eng.to_device(a, "cuda:0"))
        assert np.array_equal(eng.to_host(f), oracle.ntt(a, T, p))
        assert np.array_equal(eng.to_host(cl.inverse(f)), a)
    h = C.c_void_p()
    assert L.ntt_plan_clone(None, 0, C.byref(h)) == _lib.NTT_E_ARG
    pl = eng.NTTPlan(4, 3329, 4, 0)
    assert L.ntt_plan_clone(pl._h, 99, C.byref(h)) == _lib.NTT_E_NODEVICE
    assert L.ntt_plan_clone(pl._h, 0, None) == _lib.NTT_E_ARG
    assert L.ntt_plan_set_policy(pl._h, 5) == _lib.NTT_E_ARG and L.ntt_plan_set_policy(pl._h, -2) == _lib.NTT_E_ARG


def test_cxx_multi_device_host(tmp_path):
    """tests/cxx/multi_device_host.cpp: a C++ host shards [B][N] over ntt_device_count() devices (one plan clone and one
    stream per shard), verifies every shard the reference's way.  One device on this box, plus two extra clones on it."""
    exe = str(tmp_path / "md_host")
    lib, orc = os.path.join(ROOT, "ntt_aie_amd"), os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-C", orc, "libntt_oracle.so"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["hipcc", "-O2", "--offload-arch=gfx950", os.path.join(ROOT, "tests", "cxx", "multi_device_host.cpp"),
                           "-I" + os.path.join(ROOT, "include"), "-I" + orc, "-L" + lib, "-lntt_hip", "-L" + orc,
                           "-lntt_oracle", "-Wl,-rpath," + lib, "-Wl,-rpath," + orc, "-o", exe])
    for env_extra, shards in (({}, None), ({"NTT_MD_REPLICAS": "2"}, 3)):
        out = subprocess.run([exe, "16", "37"], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env_extra))
        assert out.returncode == 0 and "PASS!" in out.stdout and "MISMATCH" not in out.stdout, out.stdout + out.stderr
        assert "peer table copies: 0 direct (xGMI), 0 staged through the host" in out.stdout  # one device: nothing crosses a link, and the line says so
        if shards:
            assert "shards: %d" % shards in out.stdout and out.stdout.count(" ok") == shards


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` with NO launcher: the parent starts the ranks itself (torch.distributed.run as a child),
    relays rank 0's one JSON line and the exit code.  Rehearsal mode of the one-GPU box: both ranks on cuda:0, gloo."""
    env = dict(os.environ, NTT_BENCH_ONE_DEVICE="1", NTT_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--batch", "256"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and "cpu_baseline" not in d
    assert abs(d["value"] - 2 * 256 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6
    assert d["all_ranks_verified"] is True and d["world_size_seen"] == 2 and len(d["ranks"]) == 2  # round 4: every rank checked
    # config 5 is named when its per-GPU batch is asked for
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--batch", "8192"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["baseline_config"] == 5 and "config 5" in d["config"]["workload"]


def test_bench_self_launch_refuses_missing_devices():
    """More ranks than visible devices (and no rehearsal switch): a clear refusal before anything touches the GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NTT_BENCH_ONE_DEVICE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 2 and "only" in out.stderr and "device" in out.stderr and "{" not in out.stdout


P62, P64B = 0x3FFFFFEE00000001, 0xFFFFFFFC00000001  # general odd 64-bit moduli: a 62-bit NTT prime (g = 3), one above 2^63 (g = 10)


@pytest.mark.parametrize("p,g", [(P62, 3), (P64B, 10)])
def test_general_64bit_modulus(eng, oracle, p, g):
    """ntt_plan_create(..., word_bytes = 8, any odd p): FieldM64 (Montgomery, R = 2^64) behind the same passes -- BASELINE's
    metric says "64-bit prime", the reference's `%`-based network takes any modulus (src/test.cpp:48-50).  N = 2^12 (one pass)
    and 2^16 (two), every leg; tables made on the host and on the device; the product; the test_stage hook."""
    dt = np.uint64
    for logn in (12, 16):
        n = 1 << logn
        T = oracle.make_roots(n, p, g, 8)
        pl = eng.NTTPlan(logn, p, 8, 0)
        assert np.array_equal(pl.make_roots(g), T)
        pl.set_twiddles(T)
        _check_all_legs(eng, oracle, pl, T, p, dt, (1, 5), seed=logn)
        pl2 = eng.NTTPlan(logn, p, 8, 0)
        pl2.generate_twiddles(0, g)
        assert np.array_equal(pl2.get_twiddles(), T)
        a = _rand(3, n, p, dt, 77)
        a[0, :4] = np.array([0, 1, p - 1, p - 2], dtype=dt)
        f = pl2.forward(eng.to_device(a, "cuda:0"))
        assert np.array_equal(eng.to_host(f), oracle.ntt(a, T, p, nthreads=4))
        assert np.array_equal(eng.to_host(pl2.inverse(f)), a)
        for st in (0, logn // 2, logn - 1):
            got = eng.to_host(pl.forward_stages(eng.to_device(a[:1], "cuda:0"), st))
            assert np.array_equal(got, oracle.ntt(a[:1], T, p, stage=st)), st
        assert pl.count_noncanonical(eng.to_device(a, "cuda:0")) == 0
        b = a.copy()
        b[1, 7] = p
        assert pl.count_noncanonical(eng.to_device(b, "cuda:0")) == 1
    # N = 2^21 = 13 + 8: the forward 13-stage CONTIG kernel of FieldM64 on the device, every leg
    logn = 21
    T = oracle.make_roots(1 << logn, p, g, 8)
    pl = eng.NTTPlan(logn, p, 8, 0)
    pl.set_twiddles(T)
    assert [m for _, _, m in pl.passes] == [13, 8]
    _check_all_legs(eng, oracle, pl, T, p, dt, (1, 3), seed=logn)
    # negacyclic product (kind-2 table; the fused middle pass instantiated for FieldM64, as for Goldilocks) and pointwise
    for logn in (6, 12, 14, 17, 20):  # 2^6: pointwise folded into the forward pass; 2^12: ONE fused launch; 2^14 / 2^17 / 2^20: fused middle of 8 / 9 / 12 stages
        n = 1 << logn
        pl = eng.NTTPlan(logn, p, 8, 0)
        T = pl.make_table(2, g)
        assert np.array_equal(T, oracle.make_table(2, n, p, g))
        pl.set_twiddles(T)
        a, b = _rand(3, n, p, dt, 1), _rand(3, n, p, dt, 2)
        pw = eng.to_host(pl.pointwise_mul(eng.to_device(a, "cuda:0"), eng.to_device(b, "cuda:0"), scale=12345))
        assert np.array_equal(pw, oracle.pointwise(a, b, p, 12345))
        c = eng.to_host(pl.polymul_negacyclic(eng.to_device(a, "cuda:0"), eng.to_device(b, "cuda:0")))
        if logn <= 8:
            want = np.stack([oracle.negacyclic_schoolbook(a[i], b[i], p) for i in range(3)]).astype(dt)
        else:
            A, B = oracle.intt(a, T, p, nthreads=8), oracle.intt(b, T, p, nthreads=8)
            want = oracle.ntt(oracle.pointwise(A, B, p, n % p), T, p, nthreads=8)
        assert np.array_equal(c, want), logn


def test_general_64bit_modulus_any_odd(eng, oracle):
    """Not only primes: a composite odd 64-bit modulus and an arbitrary table of units (the reference never checks primality);
    2^64 - 59 (the largest 64-bit prime); a SMALL modulus in 8-byte words; even moduli are refused."""
    dt = np.uint64
    rng = np.random.default_rng(8)
    for p in (0xFFFFFFFFFFFFFFC5, (1 << 61) - 1, 3 * 5 * 17 * 257 * 65537 * 641, 3329):
        logn, n = 10, 1024
        T = (rng.integers(0, 2**63, size=n, dtype=np.uint64) % np.uint64(p)).astype(dt)
        T[T == 0] = 1
        if p == 3 * 5 * 17 * 257 * 65537 * 641:  # keep every entry a unit of the composite modulus
            T = np.array([int(t) if __import__("math").gcd(int(t), p) == 1 else 1 for t in T], dtype=dt)
        pl = eng.NTTPlan(logn, p, 8, 0)
        pl.set_twiddles(T)
        a = _rand(4, n, p, dt, 3)
        f = pl.forward(eng.to_device(a, "cuda:0"))
        assert np.array_equal(eng.to_host(f), oracle.ntt(a, T, p)), p
        assert np.array_equal(eng.to_host(pl.inverse(f)), a), p
    with pytest.raises(eng.NTTError):
        eng.NTTPlan(8, 1 << 62, 8, 0)


def test_bench_quotes_only_matching_forward_counters():
    """The headline bench line quotes hardware counters from profiles/<round>_*.json only when their kernel-source hash equals
    the tree's, and then only the FORWARD kernels' entries (INV argument false); otherwise it says why not."""
    import bench
    from ntt_aie_amd import _lib

    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu-baseline",
                          "--no-valu-floor", "--no-inverse", "--explain"],  # --explain: the full provenance strings, not the slim line's file names
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    r = d["roofline"]
    assert d["config"]["baseline_config"] == 3 and "inverse" not in d
    have, _ = bench.tagged_profile("pmc_traffic", _lib.kernel_source_hash())  # the newest committed collection on exactly these sources
    if have is not None:
        assert r["traffic"] is not None and 1.9 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 2.1  # two trips, no over-fetch
        assert "forward kernels" in r["traffic_source"] and ", true," not in r["traffic_source"].split("forward kernels:")[1].replace("true, false", "")
        v = r["valu"]
        assert v is not None and all(20 < x < 26 for x in v["instr_per_butterfly"]) and 0 < v["frac_of_peak_at_held_clock"] < 1
        assert all("false" in k.split(",")[4] for k in v["kernels"])  # PassCfg<F, LOG_M, LOG_C, CONTIG, INV, ...>: INV == false
        sa = v.get("statement_alone_steady_state")
        if sa:  # what the butterfly statement alone sustains (profiles/rNN_stream_occupancy.txt): the kernels cannot beat it
            assert 60 < sa["cycles_per_butterfly_at_4_or_more_waves"] < 100 and all(k >= 1.0 for k in sa["kernel_over_statement"])
    else:
        assert r["traffic"] is None and r["valu"] is None and ("not quoted" in r["traffic_source"] or "absent" in r["traffic_source"])
