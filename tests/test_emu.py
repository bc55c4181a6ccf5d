"""Host index model of the HIP pass kernels (same pass.h / field.h / plan.h,
compiled with g++, all 256 lanes of a workgroup stepped phase by phase) against
the oracle.  Checks the index rules, LDS exchange pattern, planner and modular
arithmetic on the CPU-only container; the GPU tests then only have to confirm
that the hardware agrees."""
import ctypes as C

import numpy as np
import pytest

import emu_lib

GOLD = 0xFFFFFFFF00000001
FIELDS = [(8, GOLD, 7), (4, 3221225473, 5), (4, 3329, 3)]


def _run(oracle, wb, logn, p, g, batch, inverse=0, layout=0, scale=1, tw=2048, ov=0, seed=0, inplace=False):
    n = 1 << logn
    dt = np.uint32 if wb == 4 else np.uint64
    T = oracle.make_roots(n, p, g, wb)
    rng = np.random.default_rng(seed)
    a = (rng.integers(0, 2**63, size=(batch, n), dtype=np.uint64) % np.uint64(p)).astype(dt)
    if not inverse:
        want = oracle.ntt(a, T, p, nthreads=4)
        if layout:
            want = oracle.block16(want)
        src = a.copy()
    else:
        src = oracle.ntt(a, T, p, nthreads=4)
        want = a.copy()
        if layout:
            src = oracle.block16(src)
        if not scale:
            want = ((want.astype(object) * n) % p).astype(dt)
    out = src if inplace else np.zeros_like(a)
    rc = emu_lib.lib().emu_transform(wb, logn, p, T.ctypes.data, src.ctypes.data, out.ctypes.data, batch,
                                     inverse, layout, scale, tw, ov)
    assert rc == 0
    assert np.array_equal(out, want)


@pytest.mark.parametrize("wb,p,g", FIELDS)
def test_single_pass_sizes(oracle, wb, p, g):
    for logn in range(1, 14 if wb == 4 else 13):  # 4-byte words: N = 2^13 is one 13-stage pass too
        for inv in (0, 1):
            for batch in (1, 19) if logn < 13 else (1, 3):
                _run(oracle, wb, logn, p, g, batch, inverse=inv, layout=int(logn >= 4 and batch != 1), seed=logn)


@pytest.mark.parametrize("wb,p,g", FIELDS[:2])
def test_multi_pass_planner_splits(oracle, wb, p, g):
    for logn in (13, 14, 16, 17, 21, 22):  # 2^21 = 13 + 8, 2^22 = 13 + 9 (the 512-row column tile)
        for inv in (0, 1):
            _run(oracle, wb, logn, p, g, 3 if logn < 21 else 1, inverse=inv, layout=1, tw=8, seed=logn, inplace=True)


@pytest.mark.parametrize("ov", [(8, 4), (8, 5), (7, 6), (8, 7), (8, 8), (9, 4), (10, 4), (11, 5), (12, 4), (5, 4), (4, 6), (5, 6), (6, 7),
                                (5, 5, 8), (13,), (13, 4), (7, 8), (9, 6), (10, 6), (8, 9), (5, 9), (14,), (14, 5)])
def test_every_tile_shape(oracle, ov):
    logn = sum(ov)
    for wb, p, g in FIELDS[:2] + [(4, 998244353, 3)]:
        if wb == 4 and ov[0] < 5:  # column tiles of 4-byte words are 32 words wide: first pass >= 5 stages
            continue
        if wb == 8 and ov[0] == 14:  # the 14-stage tile exists for 4-byte words only
            continue
        if p == 998244353 and ov[0] != 14:  # the lazy modulus class: the shape the planner actually picks the 14-stage pass for
            continue
        for inv in (0, 1):
            _run(oracle, wb, logn, p, g, 2, inverse=inv, scale=inv, tw=4, ov=emu_lib.pack_passes(*ov), seed=7)


def test_planner_covers_all_sizes():
    tri = (C.c_int * 24)()
    for wb in (4, 8):
        for logn in range(1, 29):
            k = emu_lib.lib().emu_plan(logn, wb, tri)
            passes = [(tri[3 * i], tri[3 * i + 1], tri[3 * i + 2]) for i in range(k)]
            top = 13 if wb == 4 else 12  # stages of the widest contiguous pass
            assert passes[0][0] == 1 and passes[0][1] == 0 and 1 <= passes[0][2] <= 13
            s0 = passes[0][2]
            if k > 1:
                assert s0 >= (4 if wb == 8 else 5)  # column tiles are 16 / 32 words wide
            for contig, s, m in passes[1:]:
                assert contig == 0 and s == s0 and 4 <= m <= (9 if logn == 22 else 8)
                s0 += m
            assert s0 == logn
            # fewest HBM passes (the 13-stage pass: 2^21 in two; with the 9-stage column pass 2^22 too)
            assert k == (1 if logn <= top else 2 if logn == 22 else max(2, 1 + -(-(logn - 13) // 8)))


def test_plan_alternatives():
    """plan.h: plan_alternatives / select_alternative -- the decomposition is picked at launch by batch among candidates fixed
    by (N, word size, modulus class).  Alternative 0 always exists with min_batch 0; thresholds ascend; every alternative
    covers all stages; the documented cases are present."""
    L = emu_lib.lib()
    L.emu_plan_alt.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
    L.emu_select_alt.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_uint64]
    tri, mb = (C.c_int * 24)(), C.c_uint64(0)

    def alts(logn, wb, p):
        out = []
        for a in range(8):
            k = L.emu_plan_alt(logn, wb, p, a, tri, C.byref(mb))
            if k < 0:
                break
            out.append(([tri[3 * i + 2] for i in range(k)], int(mb.value)))
            assert sum(out[-1][0]) == logn and [tri[3 * i + 1] for i in range(k)] == [sum(out[-1][0][:i]) for i in range(k)]
        return out

    for wb, p in ((8, GOLD), (4, 3221225473), (4, 998244353), (4, 3329), (4, 2013265921)):
        for logn in range(1, 29):
            a = alts(logn, wb, p)
            assert a and a[0][1] == 0 and [x[1] for x in a] == sorted(x[1] for x in a)
            for batch in (1, 2, 100, 511, 512, 1023, 1024, 4096, 1 << 20):
                k = L.emu_select_alt(logn, wb, p, batch)
                assert 0 <= k < len(a) and batch >= a[k][1] and all(batch < x[1] for x in a[k + 1:])
    g13 = alts(13, 8, GOLD)
    assert g13[0][0] == [7, 6] and g13[1][0] == [13] and g13[1][1] >= 64      # one long pass only once the batch fills the device
    assert alts(14, 4, 998244353)[1][0] == [14] and len(alts(14, 4, 3221225473)) == 1  # the 14-stage pass pays for lazy primes only
    assert alts(16, 4, 998244353)[0][0] == [10, 6] == alts(16, 4, 3221225473)[0][0]  # measured equal for the 32-bit class: one split
    # N = 2^22: two trips (13 + 9) for Goldilocks and the lazy 4-byte primes; the heavier 4-byte streams take three light passes from batch 3 on
    assert [x[0] for x in alts(22, 8, GOLD)] == [[13, 9]] == [x[0] for x in alts(22, 4, 3329)]
    assert alts(22, 4, 3221225473) == [([13, 9], 0), ([8, 7, 7], 3)] == alts(22, 4, 2013265921)
    # round 4: single-pass sizes 2^10 .. 2^12 -- the same one pass as two KERNEL variants: 512 threads x 8 words (variant 1)
    # below the batch that fills the device, the default 256 x 16 from it on; no other size has a variant
    for wb4, p4 in ((4, 3221225473), (4, 998244353), (8, GOLD), (8, 0x3FFFFFEE00000001)):
        for logn in (10, 11, 12):
            a = alts(logn, wb4, p4)
            assert [x[0] for x in a] == [[logn], [logn]] and a[0][1] == 0 and a[1][1] >= 256
            assert [L.emu_plan_alt_variant(logn, wb4, p4, k, 0) for k in (0, 1)] == [1, 0]
    assert alts(12, 4, 998244353)[1][1] <= alts(12, 4, 3221225473)[1][1]  # the lazy class hands over to radix-16 earlier
    for wb, p0, logn in ((4, 3221225473, 9), (4, 3221225473, 13), (8, GOLD, 9), (8, GOLD, 13), (4, 3221225473, 16), (8, GOLD, 16)):
        assert all(L.emu_plan_alt_variant(logn, wb, p0, k, i) == 0 for k in range(len(alts(logn, wb, p0))) for i in range(len(alts(logn, wb, p0)[k][0])))


def test_field_arithmetic_edges():
    L = emu_lib.lib()
    p = GOLD
    edge = [0, 1, 2, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFF00000000, p - 1, p - 2, 0x8000000000000000,
            0x7FFFFFFF80000001, 0xFFFFFFFE00000002]
    rng = np.random.default_rng(0)
    vals = edge + [int(x) % p for x in rng.integers(0, 2**63, size=64, dtype=np.uint64) * 2 + 1]
    for a in vals:
        for b in vals:
            assert L.emu_gl_mul(a, b) == a * b % p
            assert L.emu_gl_add(a, b) == (a + b) % p
            assert L.emu_gl_sub(a, b) == (a - b) % p
    for q in (3, 3329, 12289, 998244353, 2013265921, 3221225473, 4294967291):
        ev = [0, 1, 2, q - 1, q - 2, q // 2, q // 2 + 1]
        ev += [int(x) % q for x in rng.integers(0, 2**62, size=40)]
        for a in ev:
            for b in ev:
                assert L.emu_m32_mul_plain(a, b, q) == a * b % q
                assert L.emu_m32_add(a, b, q) == (a + b) % q
                assert L.emu_m32_sub(a, b, q) == (a - b) % q


def test_fused_pointwise_first_pass(oracle):
    """forward(in * in2 * scale) with the product folded into the first pass's load."""
    for wb, p, g in FIELDS[:2]:
        dt = np.uint32 if wb == 4 else np.uint64
        for logn in (2, 4, 6, 10, 13, 16, 17, 18):  # 13/16/17/18: radix-8 first passes of 7/8/9/10 stages
            n = 1 << logn
            T = oracle.make_roots(n, p, g, wb)
            rng = np.random.default_rng(logn)
            a = (rng.integers(0, 2**63, size=(3, n), dtype=np.uint64) % np.uint64(p)).astype(dt)
            b = (rng.integers(0, 2**63, size=(3, n), dtype=np.uint64) % np.uint64(p)).astype(dt)
            out = np.zeros_like(a)
            scale = 12345 % p
            rc = emu_lib.lib().emu_forward_product(wb, logn, p, T.ctypes.data, a.ctypes.data, b.ctypes.data,
                                                   out.ctypes.data, 3, scale, 8)
            assert rc == 0
            assert np.array_equal(out, oracle.ntt(oracle.pointwise(a, b, p, scale), T, p)), (wb, logn)


@pytest.mark.parametrize("wb,p,g,logn", [(8, GOLD, 7, l) for l in (7, 9, 10, 12, 13, 14, 16, 17, 18, 19, 20)] +
                         [(4, 998244353, 3, l) for l in (5, 6, 8, 10, 11, 12, 13, 14, 16, 17, 19, 21)] +
                         [(4, 2013265921, 31, 9), (4, 3221225473, 5, 7), (4, 3221225473, 5, 12), (4, 3221225473, 5, 14)] +
                         [(8, 0x3FFFFFEE00000001, 3, l) for l in (7, 10, 12, 14, 18)] + [(8, 0xFFFFFFFC00000001, 10, 9)])  # general 64-bit modulus
def test_product_fused_middle_pass(oracle, wb, p, g, logn):
    """The negacyclic product the way the device runs it when the first pass has a product kernel (pass.h: run_product_pass:
    last inverse pass of both operands + pointwise + first forward pass per workgroup-resident unit -- the whole product for
    N <= 2^12 --, LDS twiddle tables, register prefetch of operand b), in the host index model, against the oracle pipeline.
    Goldilocks radix-8 units of 2^7..2^12 words and 4-byte-word radix-16 units of 2^5..2^12 (lazy, 31-bit and 32-bit moduli).
    target_wgs 8 makes the workgroups stream several polynomials (the batch loop and its prefetch hand-over)."""
    dt = np.uint32 if wb == 4 else np.uint64
    n = 1 << logn
    batch = 37 if logn <= 12 else (3 if logn <= 17 else (2 if logn < 21 else 1))  # small N: several polynomials per workgroup, ragged tail
    T = oracle.make_table(2, n, p, g, wb)
    rng = np.random.default_rng(logn)
    a = (rng.integers(0, 2**63, size=(batch, n), dtype=np.uint64) % np.uint64(p)).astype(dt)
    b = (rng.integers(0, 2**63, size=(batch, n), dtype=np.uint64) % np.uint64(p)).astype(dt)
    a[0, :3] = [p - 1, 0, 1]
    b[0, :3] = [p - 1, p - 1, 0]
    A, B = oracle.intt(a, T, p, nthreads=4), oracle.intt(b, T, p, nthreads=4)
    want = oracle.ntt(oracle.pointwise(A, B, p, n % p), T, p, nthreads=4)
    sa, sb, out = a.copy(), b.copy(), np.zeros_like(a)
    rc = emu_lib.lib().emu_polymul_fused(wb, logn, p, T.ctypes.data, sa.ctypes.data, sb.ctypes.data, out.ctypes.data, batch, 8)
    assert rc == 0
    assert np.array_equal(out, want), (wb, p, logn)
    # in place into the first operand, as the Python host does by default
    sa, sb = a.copy(), b.copy()
    assert emu_lib.lib().emu_polymul_fused(wb, logn, p, T.ctypes.data, sa.ctypes.data, sb.ctypes.data, sa.ctypes.data, batch, 8) == 0
    assert np.array_equal(sa, want), (wb, p, logn)


def test_composite_odd_modulus(oracle):
    """The reference never checks that p is prime and the 4-byte-word engine takes any odd modulus: the
    forward network is the same words as the oracle's `%` arithmetic, and the inverse exists whenever every
    table entry is a unit (inverses by extended Euclid, not Fermat)."""
    p = 3 * 3329
    for logn in (6, 13):
        for inv in (0, 1):
            _run(oracle, 4, logn, p, 2, 3, inverse=inv, seed=logn)  # g = 2: every power is a unit mod 9987
    # g = 3 shares a factor with p: forward still defined, inverse refused
    n = 64
    T = oracle.make_roots(n, p, 3, 4)
    a = np.arange(n, dtype=np.uint32)[None, :]
    out = np.zeros_like(a)
    L = emu_lib.lib()
    assert L.emu_transform(4, 6, p, T.ctypes.data, a.ctypes.data, out.ctypes.data, 1, 0, 0, 1, 2048, 0) == 0
    assert np.array_equal(out, oracle.ntt(a, T, p))
    assert L.emu_transform(4, 6, p, T.ctypes.data, a.ctypes.data, out.ctypes.data, 1, 1, 0, 1, 2048, 0) == -5


def test_lds_hazard_tracker_catches_a_wrong_wave_local_rule(tmp_path):
    """The host model tracks every LDS word access (pass.h: NTT_LDS_ACCESS) and aborts when a wave touches a word that
    another wave wrote or read since the last WORKGROUP barrier.  All the tests above run under it, which is what proves
    that the exchanges the schedule declares wave-local (PassCfg::exchange_wave_local, the first hand-off of a linearly
    staged tile) really are.  Here the same model is built with a deliberately wrong rule -- every CONTIG exchange declared
    wave-local -- and must abort on a 12-stage unit, whose last exchange crosses waves."""
    import subprocess
    import sys
    import textwrap

    so = tmp_path / "libntt_emu_wrong.so"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-DNTT_EMU_FORCE_WAVE_LOCAL", emu_lib.SRC, "-o", str(so)])
    prog = textwrap.dedent("""
        import ctypes as C, sys
        import numpy as np
        L = C.CDLL(sys.argv[1])
        L.emu_transform.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint64]
        n = 1 << 12
        T = np.ones(n, dtype=np.uint32); a = np.arange(n, dtype=np.uint32).reshape(1, n); out = np.zeros_like(a)
        L.emu_transform(4, 12, 3221225473, T.ctypes.data, a.ctypes.data, out.ctypes.data, 1, 0, 0, 1, 2048, 0)
        print("no hazard reported")
    """)
    r = subprocess.run([sys.executable, "-c", prog, str(so)], capture_output=True, text=True)
    assert r.returncode != 0 and "LDS hazard" in r.stderr, (r.returncode, r.stdout, r.stderr[-500:])


def test_tapered_rows_cover_every_polynomial_group_once():
    """pass_geometry() ends a long launch with rows that stream ppw/2, ppw/4, ppw/8 polynomial groups (struct Taper): the
    row -> (first group, count) rule of phase_init() must cover every group exactly once, for ragged batches, every cap on
    ppw, several polynomials per workgroup, and the headline shape must actually be tapered."""
    L = emu_lib.lib()
    out = (C.c_uint32 * 8)()
    for batch in (1, 7, 512, 4095, 4096, 4097, 8192, 65536, 1000003):
        for cap in (1, 2, 4, 8, 64):
            for log_u in (0, 3, 5):
                for target in (8, 8192):
                    assert L.emu_geometry(16, 0, 8, 0, log_u, 1, batch, target, cap, out) == 0, (batch, cap, log_u, target, list(out))
                    assert sum(out[4:8]) == out[2]
    assert L.emu_geometry(16, 0, 8, 0, 3, 1, 4096, 8192, 8, out) == 0
    assert list(out) == [8, 32, 896, 0, 384, 128, 128, 256]  # first pass of the headline launch: 3/4 of the batch at ppw 8
    assert L.emu_geometry(16, 8, 8, 4, 0, 0, 4096, 16384, 4, out) == 0
    assert list(out)[:3] == [4, 16, 1536] and list(out)[4:] == [768, 256, 512, 0]  # its column pass: ppw 4, 2, 1
    assert L.emu_geometry(8, 0, 8, 0, 4, 1, 1 << 30, 8192, 64, out) == 0 and out[5] == 0  # beyond blockIdx.y: no taper, sliced


P62, P64B = 0x3FFFFFEE00000001, 0xFFFFFFFC00000001  # general odd 64-bit moduli (FieldM64): below 2^62 / above 2^63


def test_general_64bit_modulus_field_edges():
    """FieldM64 (Montgomery, R = 2^64) against Python integers: edge residues of a 62-bit prime, of a prime above 2^63 (sums
    carry out of the word), of Goldilocks itself taken through the general path, and of a small composite odd modulus."""
    L = emu_lib.lib()
    rng = np.random.default_rng(3)
    for p in (P62, P64B, GOLD, 0xFFFFFFFFFFFFFFC5, 3329, 3):  # 2^64 - 59: the largest 64-bit prime
        r = (1 << 64) % p
        edge = sorted({0, 1, 2 % p, p - 1, p - 2, p // 2, p // 2 + 1, (1 << 32) % p, ((1 << 32) - 1) % p, ((1 << 63) - 1) % p, (1 << 63) % p})
        vals = edge + [int(x) % p for x in rng.integers(0, 2**63, size=24, dtype=np.uint64) * 2 + 1]
        for a in vals:
            for b in vals:
                assert L.emu_m64_mul_plain(a, b, p) == a * b % p, (p, a, b)
                assert L.emu_m64_add(a, b, p) == (a + b) % p
                assert L.emu_m64_sub(a, b, p) == (a - b) % p
        # one Montgomery product with ANY 64-bit multiplicand (lazy inputs are legal for the multiplied operand)
        rinv = pow(1 << 64, -1, p)
        for x in (0, 1, p, p + 1 if p + 1 < 1 << 64 else p, (1 << 64) - 1, (1 << 63) + 12345):
            for tw in edge:
                assert L.emu_m64_mul(x, tw, p) == x * tw * rinv % p, (p, x, tw)


@pytest.mark.parametrize("p,g", [(P62, 3), (P64B, 10)])
def test_general_64bit_modulus_every_shape(oracle, p, g):
    """The pass kernels instantiated for FieldM64 through the host model: single-pass sizes, the planner's multi-pass splits,
    the 13-stage and 9-stage-column shapes, forward / inverse (scaled by phase_scale: the fold is a Goldilocks specialisation)."""
    for logn in (1, 2, 3, 4, 6, 9, 12):
        for inv in (0, 1):
            _run(oracle, 8, logn, p, g, 5, inverse=inv, layout=int(logn >= 4), seed=logn)
    for logn in (13, 16, 17):
        for inv in (0, 1):
            _run(oracle, 8, logn, p, g, 3, inverse=inv, layout=1, tw=8, seed=logn, inplace=True)
    for ov in ((13,), (8, 9), (10, 6), (5, 4), (12, 4)):
        for inv in (0, 1):
            _run(oracle, 8, sum(ov), p, g, 2, inverse=inv, scale=inv, tw=4, ov=emu_lib.pack_passes(*ov), seed=11)


@pytest.mark.parametrize("wb,p,g", [(4, 3221225473, 5), (4, 2013265921, 31), (4, 998244353, 3), (8, GOLD, 7), (8, 0x3FFFFFEE00000001, 3)])
def test_wide_radix8_variant_of_the_single_pass_sizes(oracle, wb, p, g):
    """PassDesc::variant 1 (round 4): a single-pass unit of 2^10 .. 2^12 words on 512 threads x 8 words -- radix-8 rounds of
    3 + 3 + 3 + (1..3) stages (8-byte forward: the LDS-DMA kernel, here as the LAST pass of a plan) -- instead of 256 x 16.
    Every 4-byte modulus class (32-bit, 31-bit, lazy), Goldilocks and the general 64-bit modulus, both directions (the scaled
    inverse folds N^-1 into stage 0 for 8-byte words), both layouts (the block permutation of a radix-8 round: pass.h elem_off / lane_eff), ragged batch and in
    place, under the LDS hazard tracker; 4-byte: as the first pass of a two-pass plan as well."""
    wide = 1 << 60
    for logn in (10, 11, 12):
        for inv in (0, 1):
            _run(oracle, wb, logn, p, g, 3, inverse=inv, layout=inv, scale=1, tw=4, ov=wide | emu_lib.pack_passes(logn), seed=logn)
            if logn != 11:  # in place, one polynomial, the other layout / unscaled inverse
                _run(oracle, wb, logn, p, g, 1, inverse=inv, layout=1 - inv, scale=1 - inv if inv else 1, tw=2048,
                     ov=wide | emu_lib.pack_passes(logn), seed=logn + 1, inplace=True)
    if p == 3221225473:
        for ov in ((10, 6), (12, 5)):
            for inv in (0, 1):
                _run(oracle, wb, sum(ov), p, g, 2, inverse=inv, scale=inv, tw=4, ov=wide | emu_lib.pack_passes(*ov), seed=3)
