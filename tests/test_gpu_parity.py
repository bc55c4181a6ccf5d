"""Parity of the HIP path against the oracle, through the C-ABI (libntt_hip.so),
on a real MI355X.  Bit-exact: all arithmetic is integer."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

GOLD = 0xFFFFFFFF00000001
FIELDS = [(8, GOLD, 7), (4, 3221225473, 5), (4, 998244353, 3), (4, 3329, 3)]


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


def _plan(eng, logn, p, wb, T):
    pl = eng.NTTPlan(logn, p, wb, 0)
    pl.set_twiddles(T)
    return pl


@pytest.mark.parametrize("name", sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLDEN, "*_n*.npz"))))
def test_golden_fixtures(eng, name):
    """Committed vectors: literal reference code (p <= 46340) and big-int restatement."""
    f = np.load(os.path.join(GOLDEN, name))
    n, p = int(f["n"]), int(f["p"])
    wb = 8 if p > 2**32 else 4
    dt = np.uint32 if wb == 4 else np.uint64
    T = f["table"].astype(dt)
    pl = _plan(eng, n.bit_length() - 1, p, wb, T)
    assert np.array_equal(pl.make_roots(int(f["g"])), T)  # table rule a2
    for key in [k for k in f.files if k.startswith("in_")]:
        tag = key[3:]
        a = f[key].astype(dt)[None, :]
        out = eng.to_host(pl.forward(eng.to_device(a, "cuda:0")))
        assert np.array_equal(out[0], f["out_" + tag].astype(dt)), tag
        if "out_%s_block16" % tag in f.files and n >= 16:
            out = eng.to_host(pl.forward(eng.to_device(a, "cuda:0"), layout=eng.LAYOUT_AIE_BLOCK16))
            assert np.array_equal(out[0], f["out_%s_block16" % tag].astype(dt))
    if "partial_iota" in f.files:  # test_stage hook
        a = f["in_iota"].astype(dt)[None, :]
        for s, want in enumerate(f["partial_iota"]):
            out = eng.to_host(pl.forward_stages(eng.to_device(a, "cuda:0"), s))
            assert np.array_equal(out[0], want.astype(dt)), s


@pytest.mark.parametrize("wb,p,g", FIELDS)
@pytest.mark.parametrize("logn", list(range(1, 15)) + [16, 17])
def test_forward_inverse_vs_oracle(eng, oracle, wb, p, g, logn):
    n = 1 << logn
    dt = np.uint32 if wb == 4 else np.uint64
    T = oracle.make_roots(n, p, g, wb)
    pl = _plan(eng, logn, p, wb, T)
    for batch in (1, 5, 33):
        if logn >= 16 and batch > 5:
            continue
        a = _rand(batch, n, p, dt, seed=logn * 100 + batch)
        want = oracle.ntt(a, T, p, nthreads=8)
        d = eng.to_device(a, "cuda:0")
        f = pl.forward(d)
        assert np.array_equal(eng.to_host(f), want), ("forward", batch)
        assert np.array_equal(eng.to_host(d), a), "input buffer modified"
        assert np.array_equal(eng.to_host(pl.inverse(f)), a), ("inverse", batch)
        assert np.array_equal(eng.to_host(pl.inverse(eng.to_device(want, "cuda:0"))),
                              oracle.intt(want, T, p, nthreads=8))
        unscaled = eng.to_host(pl.inverse(f, scale=False))
        assert np.array_equal(unscaled, ((a.astype(object) * n) % p).astype(dt))
        # in place, both directions
        g_ = eng.to_device(a, "cuda:0")
        pl.forward(g_, g_)
        assert np.array_equal(eng.to_host(g_), want)
        pl.inverse(g_, g_)
        assert np.array_equal(eng.to_host(g_), a)
        if logn >= 4:
            blk = pl.forward(d, layout=eng.LAYOUT_AIE_BLOCK16)
            assert np.array_equal(eng.to_host(blk), oracle.block16(want))
            assert np.array_equal(eng.to_host(pl.inverse(blk, layout=eng.LAYOUT_AIE_BLOCK16)), a)
            h_ = eng.to_device(a, "cuda:0")
            pl.forward(h_, h_, layout=eng.LAYOUT_AIE_BLOCK16)
            assert np.array_equal(eng.to_host(h_), oracle.block16(want))
        # the independent one-stage-per-launch path agrees too
        assert np.array_equal(eng.to_host(pl.forward_stages(d, logn - 1)), want)


def test_edge_values(eng, oracle):
    """All-(p-1), all-zero, alternating extremes: every carry / borrow path."""
    for wb, p, g in FIELDS:
        dt = np.uint32 if wb == 4 else np.uint64
        logn = 12
        n = 1 << logn
        T = oracle.make_roots(n, p, g, wb)
        pl = _plan(eng, logn, p, wb, T)
        rows = [np.full(n, p - 1), np.zeros(n), np.tile([0, p - 1], n // 2), np.tile([p - 1, 1, p - 2, 0], n // 4)]
        a = np.stack([np.array(r, dtype=object) for r in rows]).astype(dt)
        assert np.array_equal(eng.to_host(pl.forward(eng.to_device(a, "cuda:0"))), oracle.ntt(a, T, p))


def test_arbitrary_table_and_noninvertible(eng, oracle):
    """The kernel contract is the index rule T[h+i] only: any table of residues works;
    a table with a zero entry has no inverse and says so."""
    p, wb, logn = GOLD, 8, 10
    n = 1 << logn
    T = _rand(1, n, p, np.uint64, 99)[0]
    pl = _plan(eng, logn, p, wb, T)
    a = _rand(3, n, p, np.uint64, 5)
    f = pl.forward(eng.to_device(a, "cuda:0"))
    assert np.array_equal(eng.to_host(f), oracle.ntt(a, T, p))
    assert np.array_equal(eng.to_host(pl.inverse(f)), a)
    T[37] = 0
    pl.set_twiddles(T)
    assert not pl.has_inverse
    assert np.array_equal(eng.to_host(pl.forward(eng.to_device(a, "cuda:0"))), oracle.ntt(a, T, p))
    with pytest.raises(eng.NTTError) as ei:
        pl.inverse(f)
    assert ei.value.code == -5
    T[37] = p  # out of range
    with pytest.raises(eng.NTTError) as ei:
        pl.set_twiddles(T)
    assert ei.value.code == -7


def test_error_paths(eng):
    import torch

    pl = eng.NTTPlan(8, 3329, 4, 0)
    x = torch.zeros((2, 256), dtype=torch.int32, device="cuda:0")
    with pytest.raises(eng.NTTError) as ei:
        pl.forward(x)
    assert ei.value.code == -4  # no table yet
    pl3 = eng.NTTPlan(3, 3329, 4, 0)
    pl3.set_twiddles(pl3.make_roots(3))
    y = torch.zeros((2, 8), dtype=torch.int32, device="cuda:0")
    with pytest.raises(eng.NTTError) as ei:
        pl3.forward(y, layout=eng.LAYOUT_AIE_BLOCK16)
    assert ei.value.code == -6
    with pytest.raises(eng.NTTError):
        eng.NTTPlan(8, 3330, 4, 0)
    with pytest.raises(eng.NTTError):
        eng.NTTPlan(8, 3329, 4, 99)
    assert eng.to_host(pl3.forward(torch.zeros((0, 8), dtype=torch.int32, device="cuda:0"))).size == 0


def test_pointwise_and_negacyclic_polymul(eng, oracle):
    # 4-byte words: a lazy (< 2^30), a 31-bit and a 32-bit prime = the three butterfly streams of the product kernel (N >= 2^5)
    for wb, p, g in [(8, GOLD, 7), (4, 998244353, 3), (4, 2013265921, 31), (4, 3221225473, 5)]:
        dt = np.uint32 if wb == 4 else np.uint64
        for logn in (2, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 16):  # Goldilocks 7..12: the whole product is ONE launch of the radix-8 product kernel
            n = 1 << logn
            pl = eng.NTTPlan(logn, p, wb, 0)
            T = pl.make_table(2, g)
            assert np.array_equal(T.astype(np.uint64), oracle.make_table(2, n, p, g).astype(np.uint64))
            pl.set_twiddles(T)
            a, b = _rand(3, n, p, dt, 1), _rand(3, n, p, dt, 2)
            da, db = eng.to_device(a, "cuda:0"), eng.to_device(b, "cuda:0")
            pw = eng.to_host(pl.pointwise_mul(da, db, scale=12345))
            assert np.array_equal(pw, oracle.pointwise(a, b, p, 12345))
            out = eng.to_device(np.zeros_like(a), "cuda:0")
            c2 = eng.to_host(pl.polymul_negacyclic(eng.to_device(a, "cuda:0"), eng.to_device(b, "cuda:0"), out))
            c = eng.to_host(pl.polymul_negacyclic(da, db))  # result aliases the first operand
            assert np.array_equal(c, c2)
            if logn <= 8:
                want = np.stack([oracle.negacyclic_schoolbook(a[i], b[i], p) for i in range(3)]).astype(dt)
            else:  # oracle pipeline
                ninv = pow(n, p - 2, p)
                A, B = oracle.intt(a, T, p), oracle.intt(b, T, p)
                want = oracle.ntt(oracle.pointwise(A, B, p, n % p), T, p)
                assert ninv * n % p == 1
            assert np.array_equal(c, want), (wb, logn)


def test_full_size_properties(eng, oracle):
    """BASELINE config 3 (N = 2^16, Goldilocks, batch 4096) at full size: sampled rows
    against the oracle, round trip, and linearity (size-independent properties)."""
    import torch

    p, logn, batch = GOLD, 16, 4096
    n = 1 << logn
    pl = eng.NTTPlan(logn, p, 8, 0)
    T = pl.make_roots(7)
    pl.set_twiddles(T)
    gen = torch.Generator(device="cuda:0").manual_seed(1234)
    x = torch.randint(0, 2**62, (batch, n), dtype=torch.int64, device="cuda:0", generator=gen)  # < p: canonical
    y = torch.randint(0, 2**62, (batch, n), dtype=torch.int64, device="cuda:0", generator=gen)
    X, Y = pl.forward(x), pl.forward(y)
    rows = [0, 1, 777, 2048, 4095]
    xs = eng.to_host(x[rows])
    assert np.array_equal(eng.to_host(X[rows]), oracle.ntt(xs, T, p, nthreads=8))
    assert torch.equal(pl.inverse(X), x)
    s = x + y  # < 2^63 < p, still canonical
    S = pl.forward(s)
    Sh = eng.to_host(S[rows]).astype(object)
    assert np.array_equal(Sh, (eng.to_host(X[rows]).astype(object) + eng.to_host(Y[rows]).astype(object)) % p)
    # checksum of checksums: out[0] of every row is the plain coefficient sum
    col0 = eng.to_host(X[:64, 0]).astype(object)
    assert np.array_equal(col0, np.array([int(r.astype(object).sum()) % p for r in eng.to_host(x[:64])], dtype=object))


def test_config2_u32_batch1024(eng, oracle):
    """BASELINE config 2: N = 2^12, 32-bit prime, batch 1024 (single HBM pass)."""
    for p, g in [(12289, 11), (3221225473, 5)]:
        logn, batch = 12, 1024
        n = 1 << logn
        T = oracle.make_roots(n, p, g, 4)
        pl = _plan(eng, logn, p, 4, T)
        assert pl.hbm_passes == 1
        a = _rand(batch, n, p, np.uint32, 3)
        f = pl.forward(eng.to_device(a, "cuda:0"))
        assert np.array_equal(eng.to_host(f), oracle.ntt(a, T, p, nthreads=8))
        assert np.array_equal(eng.to_host(pl.inverse(f)), a)


def test_config4_n20_polymul_sampled(eng, oracle):
    """BASELINE config 4 shape (N = 2^20, two HBM passes) at a reduced batch."""
    p, logn, batch = GOLD, 20, 4
    n = 1 << logn
    pl = eng.NTTPlan(logn, p, 8, 0)
    T = pl.make_table(2, 7)
    pl.set_twiddles(T)
    assert pl.hbm_passes == 2
    a, b = _rand(batch, n, p, np.uint64, 11), _rand(batch, n, p, np.uint64, 12)
    f = pl.forward(eng.to_device(a, "cuda:0"))
    assert np.array_equal(eng.to_host(f[:2]), oracle.ntt(a[:2], T, p, nthreads=8))
    assert np.array_equal(eng.to_host(pl.inverse(f)), a)
    c = eng.to_host(pl.polymul_negacyclic(eng.to_device(a, "cuda:0"), eng.to_device(b, "cuda:0")))
    A, B = oracle.intt(a[:1], T, p, nthreads=8), oracle.intt(b[:1], T, p, nthreads=8)
    assert np.array_equal(c[:1], oracle.ntt(oracle.pointwise(A, B, p, n % p), T, p))


def test_reference_test_procedure(eng, oracle, tmp_path):
    """the reference's own test (src/test.cpp:137-247: logN=11, p=3329, g=3,
    a[i]=i, 10 timed launches, block-order compare, PASS/FAIL exit code) against the engine."""
    import io

    from ntt_aie_amd import host

    n, p, g = 2048, 3329, 3
    T = oracle.make_roots(n, p, g, 4)
    want = oracle.ntt(np.arange(n, dtype=np.uint32), T, p)
    assert np.array_equal(host.block_order(want), oracle.block16(want))
    buf = io.StringIO()
    rc, times = host.reference_procedure(expected_natural=want, out=buf)
    text = buf.getvalue()
    assert rc == 0 and "PASS!" in text and len(times) == 10
    assert len([l for l in text.splitlines() if l.strip().isdigit()]) == 10
    bad = want.copy()
    bad[5] ^= 1
    rc, _ = host.reference_procedure(expected_natural=bad, out=io.StringIO())
    assert rc == 1
    path = tmp_path / "ntt_mi355x_logn11.csv"
    host.write_exectime_csv(str(path), times)
    assert len(path.read_text().split()) == 10
    assert host.trimmed_mean([1996, 317, 293, 300, 310]) == pytest.approx((317 + 300 + 310) / 3)
    assert host.kerneltime_row(2048, 14.3748) == "2048 , 14.37480"
    assert host.efficiency(2048, 14.3748, 88.0) == pytest.approx(5.5 * 2048 * 11 / 14.3748e-6 / 88e9)


def test_device_generated_tables(eng, oracle):
    """tables made on the device equal the host rule, word for word, and so do
    the transforms that use them."""
    for wb, p, g in [(8, GOLD, 7), (4, 998244353, 3), (4, 3329, 3)]:
        dt = np.uint32 if wb == 4 else np.uint64
        for logn in (4, 11, 16):
            n = 1 << logn
            pl = eng.NTTPlan(logn, p, wb, 0)
            for kind in (0, 1, 2):
                try:
                    T = pl.make_table(kind, g)
                except eng.NTTError:
                    with pytest.raises(eng.NTTError):
                        pl.generate_twiddles(kind, g)
                    continue
                pl.generate_twiddles(kind, g)
                assert np.array_equal(pl.get_twiddles(), T), (wb, logn, kind)
                Ti = pl.get_twiddles(inverse=True).astype(object)
                assert all((int(x) * int(y)) % p == 1 for x, y in zip(T[1:64], Ti[1:64]))
                a = _rand(2, n, p, dt, kind)
                f = pl.forward(eng.to_device(a, "cuda:0"))
                assert np.array_equal(eng.to_host(f), oracle.ntt(a, T, p))
                assert np.array_equal(eng.to_host(pl.inverse(f)), a)


def test_graph_capture_and_streams(eng, oracle):
    """The launch functions do no allocation or synchronisation, so a transform can be captured
    into a HIP graph and replayed, and runs on whatever stream the caller hands over."""
    import torch

    p, logn, batch = GOLD, 16, 32
    n = 1 << logn
    pl = eng.NTTPlan(logn, p, 8, 0)
    T = pl.make_roots(7)
    pl.set_twiddles(T)
    a = _rand(batch, n, p, np.uint64, 77)
    x = eng.to_device(a, "cuda:0")
    y = torch.zeros_like(x)
    z = torch.zeros_like(x)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        pl.forward(x, y)  # warm-up outside capture, on a side stream
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        pl.forward(x, y)
        pl.inverse(y, z)
    y.zero_()
    z.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(eng.to_host(y[:4]), oracle.ntt(a[:4], T, p, nthreads=4))
    assert torch.equal(z, x)


def test_bench_json_contract():
    """bench.py prints one JSON line with the fields the driver reads (small batch, no CPU leg)."""
    import json
    import subprocess
    import sys

    from conftest import ROOT

    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
                          "--batch", "64", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    # round 6: the line is numbers, not commentary -- a size bound, and no clock above the part's 2.4 GHz anywhere in it
    assert len(line) < 6144, len(line)

    def leaves(o, path=""):
        if isinstance(o, dict):
            for k, v in o.items():
                yield from leaves(v, path + "/" + k)
        elif isinstance(o, list):
            for v in o:
                yield from leaves(v, path)
        else:
            yield path, o
    assert all(v <= 2.4 * 1.02 for k, v in leaves(d) if "clock" in k and "GHz" in k and isinstance(v, (int, float)))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["unit"] == "NTT/s" and d["dtype"] == "u64"
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and "workload" in d["config"]
    r = d["roofline"]
    # achieved / peak / frac: the HBM roofline of the contract (algorithmic bytes over the 8 TB/s spec peak); `bound` is decided from
    # the run's own numbers (bench.decide_bound: a roofline is named only at >= 0.9 of it), frac_ceiling is what frac can reach
    # with this pass count
    assert r["bound"] in ("hbm", "valu", "power-cap", "unsaturated") and r["roofline_of_fields"] == "hbm" and r["bound_detail"]
    assert r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 * r["frac"]  # (the default line rounds nested floats to 5 digits)
    assert "configs" not in d  # the extra configurations ride on the default shape only (batch 4096)
    assert r["frac_ceiling"] == pytest.approx(1.0 / r["passes"]) and r["frac"] <= r["frac_ceiling"]
    assert "valu" in r and "valu_source" in r and "inverse" in d and d["inverse"]["round_trip_identical"] is True


@pytest.mark.parametrize("wb,p,g", [(8, GOLD, 7), (4, 3221225473, 5), (4, 998244353, 3)])
def test_three_pass_sizes(eng, oracle, wb, p, g):
    """N = 2^21 = 13 + 8 and (round 3) 2^22 = 13 + 9 take two HBM passes; 2^23 three (CONTIG + two column passes).  For the heavy
    4-byte streams the launcher switches N = 2^22 to 8 + 7 + 7 from batch 3 on (plan alternatives): batch 2 and batch 3 both run."""
    dt = np.uint32 if wb == 4 else np.uint64
    for logn in (21, 22, 23):
        n = 1 << logn
        T = oracle.make_roots(n, p, g, wb)
        pl = _plan(eng, logn, p, wb, T)
        assert pl.hbm_passes == (3 if logn == 23 else 2)
        if logn == 22 and wb == 4 and p >= 2**30:
            assert len(pl.passes_for(2)) == 2 and len(pl.passes_for(3)) == 3
            a3 = _rand(3, n, p, dt, 5)
            f3 = pl.forward(eng.to_device(a3, "cuda:0"))
            assert np.array_equal(eng.to_host(f3), oracle.ntt(a3, T, p, nthreads=8))
            assert np.array_equal(eng.to_host(pl.inverse(f3)), a3)
        a = _rand(2, n, p, dt, logn)
        f = pl.forward(eng.to_device(a, "cuda:0"))
        assert np.array_equal(eng.to_host(f), oracle.ntt(a, T, p, nthreads=8))
        assert np.array_equal(eng.to_host(pl.inverse(f)), a)
        blk = pl.forward(eng.to_device(a, "cuda:0"), layout=eng.LAYOUT_AIE_BLOCK16)
        assert np.array_equal(eng.to_host(blk), oracle.block16(oracle.ntt(a, T, p, nthreads=8)))


def test_count_noncanonical(eng):
    for wb, p in [(8, GOLD), (4, 3329)]:
        dt = np.uint32 if wb == 4 else np.uint64
        pl = eng.NTTPlan(10, p, wb, 0)
        a = _rand(7, 1024, p, dt, 3)
        assert pl.count_noncanonical(eng.to_device(a, "cuda:0")) == 0
        a[0, 0] = p
        a[3, 77] = np.iinfo(dt).max
        a[6, 1023] = p + 1 if wb == 4 else p + 5
        assert pl.count_noncanonical(eng.to_device(a, "cuda:0")) == 3


def test_cxx_host_through_c_abi(tmp_path):
    """A C++ host (tests/cxx/ref_host.cpp, the shape of INTEGRATION.md section 1) linked against
    libntt_hip.so runs the reference's test case and verifies it the reference's way."""
    import subprocess

    from conftest import ROOT

    exe = str(tmp_path / "ref_host")
    lib, orc = os.path.join(ROOT, "ntt_aie_amd"), os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-C", orc, "libntt_oracle.so"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["hipcc", "-O2", "--offload-arch=gfx950", os.path.join(ROOT, "tests", "cxx", "ref_host.cpp"),
                           "-I" + os.path.join(ROOT, "include"), "-I" + orc, "-L" + lib, "-lntt_hip", "-L" + orc,
                           "-lntt_oracle", "-Wl,-rpath," + lib, "-Wl,-rpath," + orc, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "PASS!" in out.stdout and len([l for l in out.stdout.splitlines() if l.strip().isdigit()]) == 10


def test_bench_two_ranks_rehearsal():
    """The driver's multi-GPU launch line with two ranks, rehearsed on this one-GPU box: both ranks on
    cuda:0 and gloo instead of RCCL (which refuses two ranks on one device).  Everything else --
    table broadcast from rank 0, per-rank shards, barrier + max-over-ranks timing, the one JSON line
    of rank 0 with the whole-job aggregate -- is the code the 8-GPU run executes."""
    import json
    import subprocess
    import sys

    from conftest import ROOT

    # the launcher owns its rendezvous (--standalone: the agent binds port 0 itself and the ranks reuse its store): no port is
    # probed and handed over, a failed launch fails the test
    env = {k: v for k, v in os.environ.items() if k not in ("MASTER_ADDR", "MASTER_PORT", "RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(NTT_BENCH_ONE_DEVICE="1", NTT_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", "2",
         os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "256"],
        capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * 256 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6  # whole-job aggregate
    assert "cpu_baseline" not in d  # rank 0 at N=1 only
    # round 4: no time without a check -- every rank verified its own shard and named its device
    assert d["all_ranks_verified"] is True and d["world_size_seen"] == 2 and [r["rank"] for r in d["ranks"]] == [0, 1]
    assert all(r["round_trip_identical"] and r["coefficient_sum_invariant"] and r["ms_per_step"] > 0 for r in d["ranks"])


def test_large_sizes_against_oracle(eng, oracle):
    """N = 2^24 (8+8+8) and 2^26 (12+7+7), 4-byte words, one polynomial each: word-exact against the oracle."""
    import torch

    p, g = 3221225473, 5  # 3*2^30 + 1
    for logn in (24, 26):
        n = 1 << logn
        pl = eng.NTTPlan(logn, p, 4, 0)
        pl.generate_twiddles(0, g)  # make_roots rule, on the device
        T = pl.get_twiddles()
        assert np.array_equal(T[:4], oracle.make_roots(n, p, g, 4)[:4])
        a = _rand(1, n, p, np.uint32, logn)
        want = oracle.ntt(a, T, p, nthreads=1)
        d = eng.to_device(a, "cuda:0")
        f = pl.forward(d)
        assert np.array_equal(eng.to_host(f), want)
        assert np.array_equal(eng.to_host(pl.inverse(f)), a)
        del d, f
        torch.cuda.empty_cache()


def test_maximum_size_properties(eng):
    """N = 2^28 (the largest plan, 12+8+8), Goldilocks, 2 GiB per polynomial: size-independent properties --
    inverse(forward(a)) == a, forward(a + b) == forward(a) + forward(b), and one output word against the
    definition (the last stage's first butterfly sums everything: out[0] = sum(a) mod p)."""
    import torch

    logn, p = 28, GOLD
    n = 1 << logn
    pl = eng.NTTPlan(logn, p, 8, 0)
    pl.generate_twiddles(0, 7)
    assert pl.hbm_passes == 3
    g = torch.Generator(device="cuda:0").manual_seed(28)
    a = torch.randint(0, 1 << 62, (1, n), dtype=torch.int64, device="cuda:0", generator=g)
    b = torch.randint(0, 1 << 62, (1, n), dtype=torch.int64, device="cuda:0", generator=g)
    fa = pl.forward(a)
    assert pl.count_noncanonical(fa) == 0
    # out[0] is the plain sum of all inputs (every stage's "x + y" leg)
    ah = a.cpu().numpy().view(np.uint64).ravel()
    s = int(np.sum(ah >> np.uint64(32), dtype=np.uint64)) * (1 << 32) + int(np.sum(ah & np.uint64(0xFFFFFFFF), dtype=np.uint64))
    assert int(fa.cpu().numpy().view(np.uint64)[0, 0]) == s % p
    assert torch.equal(pl.inverse(fa), a)
    fb = pl.forward(b)
    ab = (a + b)  # < 2^63 < p
    fab = pl.forward(ab)
    # modular sum of the two transforms on the device: (fa + fb) mod p with 64-bit wraparound handled
    x, y = fa.view(torch.int64), fb.view(torch.int64)
    ssum = x + y
    carry = ((x < 0) & (y < 0)) | (((x < 0) | (y < 0)) & (ssum >= 0))  # unsigned overflow of x + y
    ssum = torch.where(carry, ssum + 0xFFFFFFFF, ssum)                # 2^64 = 2^32 - 1 (mod p)
    ge_p = (ssum < 0) & (ssum >= torch.tensor(p - (1 << 64), dtype=torch.int64, device="cuda:0"))
    ssum = torch.where(ge_p, ssum - torch.tensor(p - (1 << 64), dtype=torch.int64, device="cuda:0"), ssum)
    assert torch.equal(ssum, fab.view(torch.int64))


def test_batch_beyond_grid_limit(eng):
    """N = 2, 4-byte words, batch just above 65535 * 64 * 256 polynomials: more polynomial groups than
    blockIdx.y can number, so the launch is sliced (pass_kernel.inc:launch_cfg).  8.6 GB per buffer; the
    expected words come from the definition (src/test.cpp:46-50) evaluated with torch integer arithmetic."""
    import torch

    p, g = 3329, 3
    batch = 65535 * 64 * 256 + 777
    pl = eng.NTTPlan(1, p, 4, 0)
    T = pl.make_roots(g)
    pl.set_twiddles(T)
    gen = torch.Generator(device="cuda:0").manual_seed(5)
    a = torch.randint(0, p, (batch, 2), dtype=torch.int32, device="cuda:0", generator=gen)
    out = pl.forward(a)
    a0, a1 = a[:, 0].to(torch.int64), a[:, 1].to(torch.int64)
    assert torch.equal(out[:, 0].to(torch.int64), (a0 + a1) % p)
    assert torch.equal(out[:, 1].to(torch.int64), ((a0 - a1) % p) * int(T[1]) % p)
    del a0, a1
    assert torch.equal(pl.inverse(out), a)


def test_composite_odd_modulus(eng, oracle):
    """Any odd modulus below 2^32 (the reference never checks primality): forward == oracle; the inverse exists
    when every table entry is a unit and is refused (NTT_E_NOTINVERTIBLE) otherwise."""
    p = 3 * 3329
    for logn in (6, 13):
        n = 1 << logn
        T = oracle.make_roots(n, p, 2, 4)
        pl = _plan(eng, logn, p, 4, T)
        a = _rand(5, n, p, np.uint32, logn)
        f = pl.forward(eng.to_device(a, "cuda:0"))
        assert np.array_equal(eng.to_host(f), oracle.ntt(a, T, p))
        assert np.array_equal(eng.to_host(pl.inverse(f)), a)
    T = oracle.make_roots(64, p, 3, 4)  # powers of 3: not units mod 9987
    pl = _plan(eng, 6, p, 4, T)
    a = _rand(2, 64, p, np.uint32, 1)
    f = pl.forward(eng.to_device(a, "cuda:0"))
    assert np.array_equal(eng.to_host(f), oracle.ntt(a, T, p))
    with pytest.raises(eng.NTTError) as ei:
        pl.inverse(f)
    assert ei.value.code == -5  # NTT_E_NOTINVERTIBLE


def test_roctx_ranges_under_rocprofv3(tmp_path):
    """NTT_ROCTX=1: every transform and every HBM pass is bracketed by a ROCTX range (the role of the
    reference's trace_event0/trace_event1, src/aie_core.cc:129-131); rocprofv3 --marker-trace must list them."""
    import shutil
    import subprocess

    from conftest import ROOT

    if shutil.which("rocprofv3") is None:
        pytest.skip("rocprofv3 not on PATH")
    script = tmp_path / "roctx_run.py"
    script.write_text(
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np, torch\n"
        "from ntt_aie_amd import NTTPlan, to_device\n"
        "pl = NTTPlan(16, 0xFFFFFFFF00000001, 8, 0); pl.generate_twiddles(0, 7)\n"
        "x = to_device(np.arange(1 << 16, dtype=np.uint64)[None, :], 'cuda:0')\n"
        "y = pl.inverse(pl.forward(x)); torch.cuda.synchronize(); assert torch.equal(x, y)\n" % ROOT)
    out = tmp_path / "prof"
    env = dict(os.environ, NTT_ROCTX="1", TMPDIR="/tmp")
    r = subprocess.run(["rocprofv3", "--marker-trace", "--kernel-trace", "--output-format", "csv", "-d", str(out), "-o", "run",
                        "--", "python3", str(script)], capture_output=True, text=True, timeout=600, env=env, cwd="/tmp")
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    files = glob.glob(str(out / "**" / "*.csv"), recursive=True)
    text = "".join(open(f).read() for f in files if "marker" in os.path.basename(f))
    assert text, (files, (r.stdout + r.stderr)[-2000:])
    import ntt_aie_amd as E

    (_, _, m0), (_, s1, m1) = E.NTTPlan(16, GOLD, 8, 0).passes  # the planner's split of N = 2^16
    for name in ("ntt_forward", "fwd pass contig stages 0-%d" % (m0 - 1), "fwd pass column stages %d-%d" % (s1, s1 + m1 - 1),
                 "ntt_inverse", "inv pass column stages %d-%d" % (s1, s1 + m1 - 1), "inv pass contig stages 0-%d" % (m0 - 1)):
        assert name in text, name


def test_randomized_differential(eng, oracle):
    """Seeded random sweep over moduli (tiny, around 2^31, up to 2^32-1, composite odd, Goldilocks, general 64-bit), sizes, batches,
    layouts, in-place / out-of-place and both directions: every output word equals the oracle's.  The table rule is
    make_roots with its integer division (src/test.cpp:28), so the moduli need not be NTT-friendly."""
    moduli4 = [3, 5, 7, 3329, 12289, 40961, 65537, 786433, 8380417, 469762049, 998244353, 2013265921, 2147483647,
               2147483649, 2147483659, 3221225473, 4293918721, 4294967291, 4294967295]
    # 8-byte words: Goldilocks (fast path) and general odd moduli (round 3: FieldM64) -- tiny, 33-bit, 62-bit, just below / above 2^63,
    # the largest 64-bit prime, a composite, 2^64 - 1
    moduli8 = [GOLD, GOLD, 3, 4294967311, 0x3FFFFFEE00000001, (1 << 63) - 25, (1 << 63) + 29, 0xFFFFFFFC00000001, 0xFFFFFFFFFFFFFFC5,
               3 * 5 * 17 * 257 * 65537 * 641, (1 << 64) - 1]
    rng = np.random.default_rng(20261003)
    for case in range(int(os.environ.get("NTT_TEST_RANDOM_CASES", "96"))):  # soak: NTT_TEST_RANDOM_CASES=2000
        wb = 8 if case % 3 == 2 else 4
        p = int(moduli8[int(rng.integers(len(moduli8)))]) if wb == 8 else int(moduli4[int(rng.integers(len(moduli4)))])
        g = int(rng.integers(2, 50))
        logn = int(rng.integers(1, 18 if case >= 72 else 16))
        n = 1 << logn
        batch = int(rng.integers(1, 38))
        layout = int(rng.integers(2)) if logn >= 4 else 0
        dt = np.uint32 if wb == 4 else np.uint64
        T = oracle.make_roots(n, p, g, wb)
        pl = _plan(eng, logn, p, wb, T)
        nalt = len(pl.alternatives)
        if nalt > 1 and case % 2:  # pin a non-default decomposition now and then (plan alternatives compute the same words)
            pl.set_policy(int(rng.integers(nalt)))
        a = _rand(batch, n, p, dt, case)
        want = oracle.ntt(a, T, p, nthreads=4)
        d = eng.to_device(a, "cuda:0")
        f = pl.forward(d, d if case % 2 else None, layout=layout)  # odd cases transform in place
        got = eng.to_host(f)
        assert np.array_equal(got, oracle.block16(want) if layout else want), (case, wb, p, g, logn, batch, layout)
        if pl.has_inverse:
            back = pl.inverse(f, f if case % 3 == 0 else None, layout=layout)
            assert np.array_equal(eng.to_host(back), a), (case, wb, p, g, logn, batch, layout)


def test_against_literal_reference_library(eng, oracle):
    """HIP path vs the COMPILED LITERAL reference lines (oracle/_ref/libntt_ref.so: src/test.cpp:15-60, 69-71,
    212-219 built by oracle/build_ref.sh; the prebuilt library travels to the GPU box), inside the literal code's
    validity window p <= 46340: table, full network, intermediate stages (test_stage hook) and block order."""
    if not oracle.have_ref():
        pytest.skip("literal reference library not present on this box")
    rng = np.random.default_rng(7)
    for n, p, g in [(16, 3329, 3), (256, 3329, 3), (2048, 3329, 3), (4096, 12289, 11), (8192, 40961, 3), (1 << 14, 46337, 5)]:
        logn = n.bit_length() - 1
        Tr = oracle.ref_make_roots(n, p, g)
        pl = eng.NTTPlan(logn, p, 4, 0)
        assert np.array_equal(pl.make_roots(g).astype(np.int64), Tr.astype(np.int64))
        pl.set_twiddles(Tr.astype(np.uint32))
        a = rng.integers(0, p, size=n, dtype=np.int64)
        d = eng.to_device(a.astype(np.uint32)[None, :], "cuda:0")
        want = oracle.ref_ntt(a.astype(np.int32), Tr, p, logn - 1)
        assert np.array_equal(eng.to_host(pl.forward(d))[0].astype(np.int64), want.astype(np.int64)), (n, p)
        blk = eng.to_host(pl.forward(d, layout=eng.LAYOUT_AIE_BLOCK16))[0]
        assert np.array_equal(blk.astype(np.int32), oracle.ref_block_order(want)), (n, p)
        for stage in {0, logn // 2}:
            got = eng.to_host(pl.forward_stages(d, stage))[0]
            assert np.array_equal(got.astype(np.int64), oracle.ref_ntt(a.astype(np.int32), Tr, p, stage).astype(np.int64))


def test_lazy_range_edges_small_moduli(eng, oracle):
    """p < 2^30 runs the lazy butterflies (values in [0, 2p) between stages, canonicalised once at the end):
    extreme residues and extreme twiddles at moduli on both sides of the 2^30 and 2^31 switch points, single- and
    multi-pass sizes, forward, scaled and unscaled inverse -- word-exact against the oracle."""
    for p in (3, 5, 3329, 12289, 1073741789, 1073741823, 1073741825, 1073741827, 2147483647, 2147483649):
        for logn in (3, 5, 8, 12, 13):
            n = 1 << logn
            rng = np.random.default_rng(p % 1000 + logn)
            tables = [np.full(n, p - 1, dtype=np.uint32), oracle.make_roots(n, p, 2, 4),
                      (rng.integers(1, p, size=n, dtype=np.int64)).astype(np.uint32)]
            rows = [np.full(n, p - 1), np.zeros(n), np.tile([p - 1, 0], n // 2), np.tile([0, p - 1], n // 2),
                    rng.integers(0, p, size=n), np.full(n, p // 2), np.full(n, 1)]
            a = np.stack(rows).astype(np.uint32)
            for T in tables:
                pl = _plan(eng, logn, p, 4, T)
                want = oracle.ntt(a, T, p)
                d = eng.to_device(a, "cuda:0")
                f = pl.forward(d)
                assert np.array_equal(eng.to_host(f), want), (p, logn)
                assert pl.count_noncanonical(f) == 0
                if pl.has_inverse:
                    assert np.array_equal(eng.to_host(pl.inverse(f)), a), (p, logn)
                    un = eng.to_host(pl.inverse(f, scale=False))  # N * a mod p, canonical
                    assert np.array_equal(un, ((a.astype(np.uint64) * np.uint64(n % p)) % np.uint64(p)).astype(np.uint32)), (p, logn)


def test_goldilocks_inverse_lazy_representatives(eng, oracle):
    """The Goldilocks inverse butterflies carry sums as arbitrary 64-bit representatives; outputs must still be
    canonical and exact: extreme residues / twiddles, scaled and unscaled inverse, single-, two- and three-pass sizes."""
    p = GOLD
    for logn in (2, 5, 12, 13, 16, 21):
        n = 1 << logn
        rng = np.random.default_rng(logn)
        tables = [oracle.make_roots(n, p, 7, 8), np.full(n, p - 1, dtype=np.uint64)]
        rows = [np.full(n, p - 1, dtype=np.uint64), np.zeros(n, dtype=np.uint64),
                np.tile(np.array([p - 1, 0], dtype=np.uint64), n // 2), np.full(n, 0xFFFFFFFF, dtype=np.uint64),
                np.full(n, 0xFFFFFFFF00000000, dtype=np.uint64), _rand(1, n, p, np.uint64, logn)[0]]
        a = np.stack(rows)
        for T in tables:
            pl = _plan(eng, logn, p, 8, T)
            y = oracle.ntt(a, T, p, nthreads=4)
            d = eng.to_device(y, "cuda:0")
            back = pl.inverse(d)
            assert pl.count_noncanonical(back) == 0
            assert np.array_equal(eng.to_host(back), a), logn
            un = pl.inverse(d, scale=False)
            assert pl.count_noncanonical(un) == 0
            want = np.array([[(int(v) * n) % p for v in row] for row in a[:, :64]], dtype=np.uint64)
            assert np.array_equal(eng.to_host(un)[:, :64], want), logn


def test_shared_plan_from_two_host_threads_and_streams(eng, oracle):
    """include/ntt_hip.h: a plan is immutable after set_twiddles and may be shared by host threads.  Two threads drive the
    same plan on their own streams (forward, inverse, count_noncanonical -- the one entry point that writes plan state, behind
    a mutex) while the main thread checks every result against the oracle."""
    import threading

    import torch

    p, logn, batch = GOLD, 14, 24
    n = 1 << logn
    pl = eng.NTTPlan(logn, p, 8, 0)
    T = pl.make_roots(7)
    pl.set_twiddles(T)
    data = [_rand(batch, n, p, np.uint64, 900 + i) for i in range(2)]
    want = [oracle.ntt(a, T, p, nthreads=4) for a in data]
    results, errors = [None, None], []

    def worker(i):
        try:
            torch.cuda.set_device(0)
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                x = eng.to_device(data[i], "cuda:0")
                for _ in range(25):
                    f = pl.forward(x, stream=s)
                    back = pl.inverse(f, stream=s)
                    s.synchronize()
                    assert pl.count_noncanonical(f) == 0
                    assert torch.equal(back, x)
                results[i] = eng.to_host(f)
        except Exception as e:  # surfaced in the main thread
            errors.append(repr(e))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    for i in range(2):
        assert np.array_equal(results[i], want[i])
