"""Pin the CPU restatement (oracle/ntt_oracle.c) before anything trusts it.

Sources of truth, strongest first:
  * oracle/_ref/libntt_ref.so -- the literal reference lines compiled where they
    lie (only in the authoring container; skipped when absent),
  * tests/golden/literal_*.npz -- outputs of that literal code, committed,
  * tests/golden/bigint_*.npz -- independent Python big-int restatement, for
    primes outside the literal int32 window,
  * the words SURVEY.md section 8(c) lists (T[1], out[0..7], ...).
"""
import glob
import math
import os

import numpy as np
import pytest

from conftest import GOLDEN

GOLD = 0xFFFFFFFF00000001


def _load(name):
    return np.load(os.path.join(GOLDEN, name))


LITERAL = sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLDEN, "literal_n*.npz")))
BIGINT = sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLDEN, "bigint_n*.npz")))


def test_fixtures_present():
    assert len(LITERAL) == 5 and len(BIGINT) == 9


@pytest.mark.parametrize("name", LITERAL)
def test_restatement_vs_literal_fixtures(oracle, name):
    f = _load(name)
    n, p, g = int(f["n"]), int(f["p"]), int(f["g"])
    T = oracle.make_roots(n, p, g, 4)
    assert np.array_equal(T.astype(np.int64), f["table"].astype(np.int64))
    for tag in ("iota", "rand"):
        out = oracle.ntt(f["in_" + tag].astype(np.uint32), T, p)
        assert np.array_equal(out.astype(np.int64), f["out_" + tag].astype(np.int64)), tag
    blk = oracle.block16(oracle.ntt(f["in_iota"].astype(np.uint32), T, p))
    assert np.array_equal(blk.astype(np.int64), f["out_iota_block16"].astype(np.int64))
    if "partial_iota" in f.files:  # test_stage hook, test.cpp:55-58
        for s, want in enumerate(f["partial_iota"]):
            got = oracle.ntt(f["in_iota"].astype(np.uint32), T, p, stage=s)
            assert np.array_equal(got.astype(np.int64), want.astype(np.int64)), s


@pytest.mark.parametrize("name", BIGINT)
def test_restatement_vs_bigint_fixtures(oracle, name):
    f = _load(name)
    n, p, g = int(f["n"]), int(f["p"]), int(f["g"])
    wb = f["table"].dtype.itemsize
    T = oracle.make_roots(n, p, g, wb)
    assert np.array_equal(T, f["table"])
    for tag in ("rand", "edge"):
        assert np.array_equal(oracle.ntt(f["in_" + tag], T, p), f["out_" + tag]), tag


def test_survey_kat_words(oracle):
    """the words it lists for the literal code (its FNV digests are
    not reproducible from its description, the listed words are)."""
    kats = {
        (16, 3329, 3): dict(T={1: 2699}, head=[120, 153, 3260, 1346, 1080, 2105, 52, 654,
                                                372, 1912, 1654, 3027, 780, 998, 1322, 1892]),
        (256, 3329, 3): dict(T={1: 3061, 2: 1915, 128: 3328, 255: 2298},
                             head=[2679, 3131, 2635, 2619, 2323, 1139, 1135, 2694], tail=[2707, 2956]),
        (2048, 3329, 3): dict(T={1: 3, 1024: 341, 2047: 3251},
                              head=[2187, 1952, 747, 1368, 1399, 3021, 3063, 854], tail=[3043, 1667]),
        (4096, 12289, 11): dict(T={1: 1331, 2: 1945, 2048: 12288},
                                head=[5462, 7281, 3146, 2408, 5459, 7072, 4252, 5089], tail=[7891, 6003]),
        (8192, 40961, 3): dict(T={1: 243}, head=[3277, 24712, 18859, 36922, 16028, 15594, 35444, 29008]),
    }
    for (n, p, g), k in kats.items():
        T = oracle.make_roots(n, p, g, 4)
        for i, v in k["T"].items():
            assert int(T[i]) == v
        out = oracle.ntt((np.arange(n) % p).astype(np.uint32), T, p)
        assert out[: len(k["head"])].tolist() == k["head"]
        if "tail" in k:
            assert out[-2:].tolist() == k["tail"]


def test_restatement_vs_literal_library(oracle):
    """Direct word-for-word comparison against the compiled literal lines."""
    if not oracle.have_ref():
        pytest.skip("reference tree absent: literal library not built")
    rng = np.random.default_rng(11)
    for n, p, g in [(2, 3329, 3), (4, 3329, 3), (8, 3329, 3), (32, 3329, 3), (512, 12289, 11),
                    (1024, 40961, 3), (2048, 3329, 3), (4096, 46337, 5)]:
        logn = n.bit_length() - 1
        T = oracle.make_roots(n, p, g, 4)
        Tr = oracle.ref_make_roots(n, p, g)
        assert np.array_equal(T.astype(np.int64), Tr.astype(np.int64))
        a = rng.integers(0, p, size=n, dtype=np.int64)
        for stage in {0, logn // 2, logn - 1}:
            got = oracle.ntt(a.astype(np.uint32), T, p, stage=stage)
            want = oracle.ref_ntt(a.astype(np.int32), Tr, p, stage)
            assert np.array_equal(got.astype(np.int64), want.astype(np.int64)), (n, p, stage)
    a = np.arange(2048, dtype=np.int32)
    assert np.array_equal(oracle.block16(a.astype(np.uint32)).astype(np.int32), oracle.ref_block_order(a))


def test_scalar_twins_fixture():
    """aie_core.cc:11-39 (modadd/modsub/barrett_2k) equal plain modular arithmetic
    on the committed samples, with aie2.py:17-19's constants."""
    f = _load("literal_scalar_q3329.npz")
    q, w, u = int(f["q"]), int(f["w"]), int(f["u"])
    assert w == math.ceil(math.log2(q)) and u == (1 << (2 * w)) // q
    a, b = f["ab"][:, 0].astype(np.int64), f["ab"][:, 1].astype(np.int64)
    assert np.array_equal(f["modadd"], (a + b) % q)
    assert np.array_equal(f["modsub"], (a - b) % q)
    assert np.array_equal(f["barrett"], (a * b) % q)


@pytest.mark.parametrize("wb,p,g", [(4, 12289, 11), (4, 998244353, 3), (4, 3221225473, 5), (8, GOLD, 7)])
def test_inverse_roundtrip_and_linearity(oracle, wb, p, g):
    n = 512
    T = oracle.make_roots(n, p, g, wb)
    rng = np.random.default_rng(5)
    dt = T.dtype
    a = (rng.integers(0, 2**63, size=(4, n), dtype=np.uint64) % np.uint64(p)).astype(dt)
    b = (rng.integers(0, 2**63, size=(4, n), dtype=np.uint64) % np.uint64(p)).astype(dt)
    A, B = oracle.ntt(a, T, p), oracle.ntt(b, T, p)
    assert np.array_equal(oracle.intt(A, T, p), a)
    s = ((a.astype(object) + b.astype(object)) % p).astype(dt)
    S = ((A.astype(object) + B.astype(object)) % p).astype(dt)
    assert np.array_equal(oracle.ntt(s, T, p), S)
    # multi-threaded driver == scalar driver
    assert np.array_equal(oracle.ntt(a, T, p, nthreads=4), A)


def test_edge_sizes(oracle):
    p = GOLD
    for n in (2, 4):
        T = oracle.make_roots(n, p, 7, 8)
        a = np.array([p - 1] * n, dtype=np.uint64)
        out = oracle.ntt(a, T, p)
        assert out[0] == (n * (p - 1)) % p
        assert np.array_equal(oracle.intt(out, T, p), a)


def test_f6_tables(oracle):
    """(i) cyclic table => DFT of the bit-reversed input;
    (ii) Longa-Naehrig psi^-1 table => unscaled negacyclic inverse NTT."""
    p, n = GOLD, 32
    logn = 5
    w = pow(7, (p - 1) // n, p)
    T1 = oracle.make_table(1, n, p, 7)
    rng = np.random.default_rng(3)
    a = (rng.integers(0, 2**63, size=n, dtype=np.uint64) % np.uint64(p))
    out = oracle.ntt(a, T1, p)
    brv = [int(format(j, "0%db" % logn)[::-1], 2) for j in range(n)]
    for k in range(n):
        assert int(out[k]) == sum(int(a[brv[j]]) * pow(w, j * k, p) for j in range(n)) % p
    T2 = oracle.make_table(2, n, p, 7)
    b = (rng.integers(0, 2**63, size=n, dtype=np.uint64) % np.uint64(p))
    A, B = oracle.intt(a, T2, p), oracle.intt(b, T2, p)
    c = oracle.ntt(oracle.pointwise(A, B, p, n % p), T2, p)
    assert np.array_equal(c, oracle.negacyclic_schoolbook(a, b, p))
