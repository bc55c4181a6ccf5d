"""Parity at the REAL batch of every BASELINE.json configuration that fits one GPU, and of every pass kernel
with a multi-iteration batch loop (ppw >= 2: counted-vmcnt LDS-DMA hand-off, double-buffered tiles).

The batch changes `ppw`, `grid_y` and the loop counts of the pass kernels (pass.h:pass_geometry), which is
exactly where an indexing bug would hide, so these cases run the C-ABI at full size and compare sampled rows
word for word with the oracle plus size-independent properties over the WHOLE batch (round trip, coefficient
sum).  Reference analogue of the word-exact check: src/test.cpp:203-235."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = 0xFFFFFFFF00000001


@pytest.fixture(scope="module")
def eng():
    import torch

    import ntt_aie_amd as E

    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    return E


def _device_batch(torch, batch, n, p, wb, seed):
    """Canonical residues generated on the device (host generation of 4 GiB would dominate the test)."""
    g = torch.Generator(device="cuda:0").manual_seed(seed)
    if wb == 8:
        return torch.randint(0, 2**62, (batch, n), dtype=torch.int64, device="cuda:0", generator=g)  # < p
    return torch.randint(0, p, (batch, n), dtype=torch.int64, device="cuda:0", generator=g).to(torch.int32)


def _rowsum_mod_p(rows: np.ndarray, p: int) -> list:
    return [int(r.astype(object).sum()) % p for r in rows]


def test_config4_n20_batch512_forward_inverse_polymul(eng, oracle):
    """BASELINE config 4 at its real batch: N = 2^20, Goldilocks, negacyclic (kind-2) table, batch 512.
    ppw = 16 in the 512-thread LDS-DMA first pass, 8 in the column pass."""
    import torch

    p, logn, batch = GOLD, 20, 512
    n = 1 << logn
    pl = eng.NTTPlan(logn, p, 8, 0)
    T = pl.make_table(2, 7)
    pl.set_twiddles(T)
    assert pl.hbm_passes == 2
    rows = [0, 1, 255, 256, 511]
    a = _device_batch(torch, batch, n, p, 8, 2004)
    ah = eng.to_host(a[rows])
    f = pl.forward(a)
    assert np.array_equal(eng.to_host(f[rows]), oracle.ntt(ah, T, p, nthreads=8)), "forward rows"
    assert pl.count_noncanonical(f) == 0
    # out[0] of every polynomial is its plain coefficient sum (every stage's x + y leg): the whole batch
    col0 = eng.to_host(f[:, 0].contiguous())
    sample = list(range(0, batch, 37))
    assert [int(v) for v in col0[sample]] == _rowsum_mod_p(eng.to_host(a[sample]), p)
    back = pl.inverse(f)
    assert torch.equal(back, a), "inverse(forward) over the whole batch"
    del back, f
    # the product: rows {0, 511} against the oracle pipeline (unscaled inverse network on both operands,
    # pointwise * N^-1, forward network: SURVEY F6-ii), plus linearity in the first operand over the whole batch
    b = _device_batch(torch, batch, n, p, 8, 2005)
    prow = [0, 511]
    ah2, bh2 = eng.to_host(a[prow]), eng.to_host(b[prow])
    A, B = oracle.intt(ah2, T, p, nthreads=8), oracle.intt(bh2, T, p, nthreads=8)
    want = oracle.ntt(oracle.pointwise(A, B, p, n % p), T, p, nthreads=8)
    a2, b2 = a.clone(), b.clone()
    c = pl.polymul_negacyclic(a2, b2)  # operands are scratch; result aliases a2
    assert np.array_equal(eng.to_host(c[prow]), want), "polymul rows"
    assert pl.count_noncanonical(c) == 0
    # c[i][0] = a[i][0]*b[i][0] - sum_{j>=1} a[i][j]*b[i][N-j]: too costly for 512 rows on the host; instead the
    # product with b = 1 (the constant polynomial) must return a, on the whole batch
    one = torch.zeros_like(b)
    one[:, 0] = 1
    a3 = a.clone()
    c1 = pl.polymul_negacyclic(a3, one)
    assert torch.equal(c1, a), "a * 1 over the whole batch"


def test_config5_per_gpu_shape_n16_batch8192(eng, oracle):
    """BASELINE config 5's per-GPU shard: N = 2^16, Goldilocks, batch 8192 (4 GiB per buffer)."""
    import torch

    p, logn, batch = GOLD, 16, 8192
    n = 1 << logn
    pl = eng.NTTPlan(logn, p, 8, 0)
    T = pl.make_roots(7)
    pl.set_twiddles(T)
    x = _device_batch(torch, batch, n, p, 8, 5005)
    X = pl.forward(x)
    rows = [0, 1, 4095, 4096, 4097, 6000, 8190, 8191]
    assert np.array_equal(eng.to_host(X[rows]), oracle.ntt(eng.to_host(x[rows]), T, p, nthreads=8))
    assert pl.count_noncanonical(X) == 0
    sample = list(range(0, batch, 257))
    col0 = eng.to_host(X[:, 0].contiguous())
    assert [int(v) for v in col0[sample]] == _rowsum_mod_p(eng.to_host(x[sample]), p)  # out[0] = sum(a)
    assert torch.equal(pl.inverse(X), x)
    # in place at the same batch
    y = x.clone()
    pl.forward(y, y)
    assert torch.equal(y, X)


@pytest.mark.parametrize("logn,batch", [(13, 8192), (14, 4096), (15, 2048), (17, 1024), (18, 512), (19, 256)])
def test_multi_iteration_batch_loops_goldilocks(eng, oracle, logn, batch):
    """Every Goldilocks first-pass kernel (256-thread LDS-DMA radix-8 for 7-9 stages, 512-thread one for 10-12) and
    its non-DMA twin with the fused pointwise product, at a batch that makes the workgroups stream several polynomials
    (ppw >= 4): forward, inverse, product -- sampled rows against the oracle, whole batch by round trip."""
    import torch

    p = GOLD
    n = 1 << logn
    pl = eng.NTTPlan(logn, p, 8, 0)
    T = pl.make_table(2, 7)
    pl.set_twiddles(T)
    a = _device_batch(torch, batch, n, p, 8, logn)
    b = _device_batch(torch, batch, n, p, 8, logn + 100)
    rows = [0, 1, batch // 2 - 1, batch // 2, batch - 2, batch - 1]
    ah, bh = eng.to_host(a[rows]), eng.to_host(b[rows])
    f = pl.forward(a)
    assert np.array_equal(eng.to_host(f[rows]), oracle.ntt(ah, T, p, nthreads=8))
    assert torch.equal(pl.inverse(f), a)
    g = pl.inverse(a, scale=False)
    assert np.array_equal(eng.to_host(g[rows]), oracle.pointwise(oracle.intt(ah, T, p, nthreads=8), np.ones_like(ah), p, n % p))
    A, B = oracle.intt(ah, T, p, nthreads=8), oracle.intt(bh, T, p, nthreads=8)
    want = oracle.ntt(oracle.pointwise(A, B, p, n % p), T, p, nthreads=8)
    c = pl.polymul_negacyclic(a.clone(), b.clone())
    assert np.array_equal(eng.to_host(c[rows]), want)
    # the operands as ONE [2*batch][N] buffer: the library transforms both with one launch per pass
    ab = torch.cat([a, b])
    c2 = pl.polymul_negacyclic(ab[:batch], ab[batch:])
    assert c2.data_ptr() == ab.data_ptr() and torch.equal(c2, c)
    del ab, c2
    one = torch.zeros_like(b)
    one[:, 0] = 1
    assert torch.equal(pl.polymul_negacyclic(a.clone(), one), a)


@pytest.mark.parametrize("p,g", [(998244353, 3), (3221225473, 5)])
@pytest.mark.parametrize("logn,batch", [(12, 65536), (13, 32771), (16, 8192), (20, 256), (21, 131)])
def test_multi_iteration_batch_loops_u32(eng, oracle, p, g, logn, batch):
    """4-byte words (lazy p < 2^30 and carry-select p >= 2^31 streams) at batches with ppw >= 2 in every pass; 2^13 is the
    13-stage contiguous pass alone (512 threads x 16 words, 8192-word tile) and 2^21 the same pass + an 8-stage column pass."""
    assert len(eng.NTTPlan(13, p, 4, 0).passes) == 1 and len(eng.NTTPlan(21, p, 4, 0).passes) == 2
    import torch

    n = 1 << logn
    T = oracle.make_roots(n, p, g, 4)
    pl = eng.NTTPlan(logn, p, 4, 0)
    pl.set_twiddles(T)
    a = _device_batch(torch, batch, n, p, 4, logn)
    rows = [0, 1, batch // 2, batch - 1]
    f = pl.forward(a)
    assert np.array_equal(eng.to_host(f[rows]), oracle.ntt(eng.to_host(a[rows]), T, p, nthreads=4))
    assert pl.count_noncanonical(f) == 0
    assert torch.equal(pl.inverse(f), a)


@pytest.mark.parametrize("logn,batch", [(13, 5), (16, 3), (21, 2), (22, 1)])
def test_product_fused_middle_aliasing_and_three_pass(eng, oracle, logn, batch):
    """The fused middle pass of the product (pass.h:run_product_pass) at odd batches, for a three-pass size (N = 2^22:
    two inverse column passes, the fused middle, two forward column passes), for N = 2^21 = 13 + 8 stages (no product kernel
    for the 13-stage 8-byte unit: the pointwise leg is folded into the forward pass instead), and with the result buffer
    aliasing either operand or neither -- all must give the oracle pipeline's words."""
    import torch

    p = GOLD
    n = 1 << logn
    pl = eng.NTTPlan(logn, p, 8, 0)
    T = pl.make_table(2, 7)
    pl.set_twiddles(T)
    rng = np.random.default_rng(logn)
    a = (rng.integers(0, 2**63, size=(batch, n), dtype=np.uint64) % np.uint64(p))
    b = (rng.integers(0, 2**63, size=(batch, n), dtype=np.uint64) % np.uint64(p))
    a[0, :4] = [p - 1, 0, p - 1, 1]  # edge residues in the first unit
    b[0, :4] = [p - 1, p - 1, 0, 2]
    A, B = oracle.intt(a, T, p, nthreads=8), oracle.intt(b, T, p, nthreads=8)
    want = oracle.ntt(oracle.pointwise(A, B, p, n % p), T, p, nthreads=8)
    da, db = eng.to_device(a, "cuda:0"), eng.to_device(b, "cuda:0")
    c = pl.polymul_negacyclic(da, db)  # result aliases a
    assert c.data_ptr() == da.data_ptr() and np.array_equal(eng.to_host(c), want)
    assert pl.count_noncanonical(c) == 0
    da, db = eng.to_device(a, "cuda:0"), eng.to_device(b, "cuda:0")
    c = pl.polymul_negacyclic(da, db, db)  # result aliases b
    assert np.array_equal(eng.to_host(c), want)
    da, db = eng.to_device(a, "cuda:0"), eng.to_device(b, "cuda:0")
    out = torch.zeros_like(da)
    assert np.array_equal(eng.to_host(pl.polymul_negacyclic(da, db, out)), want)
    # commutativity on the device: b * a gives the same words
    da, db = eng.to_device(a, "cuda:0"), eng.to_device(b, "cuda:0")
    assert np.array_equal(eng.to_host(pl.polymul_negacyclic(db, da)), want)


@pytest.mark.parametrize("logn,batch", [(7, 100003), (10, 16411), (12, 8192)])
def test_single_pass_product_one_launch(eng, oracle, logn, batch):
    """Goldilocks, 2^7 <= N <= 2^12: the whole negacyclic product is ONE launch of the product kernel (inverse network on
    a, on b, pointwise, forward network; several polynomials per workgroup below N = 2^11, ragged batch tail): sampled
    rows against the oracle pipeline and against the schoolbook product, a * 1 = a over the whole batch."""
    import torch

    p = GOLD
    n = 1 << logn
    pl = eng.NTTPlan(logn, p, 8, 0)
    T = pl.make_table(2, 7)
    pl.set_twiddles(T)
    assert pl.hbm_passes == 1
    a = _device_batch(torch, batch, n, p, 8, logn)
    b = _device_batch(torch, batch, n, p, 8, logn + 50)
    rows = [0, 1, 2, batch // 2, batch - 3, batch - 2, batch - 1]
    ah, bh = eng.to_host(a[rows]), eng.to_host(b[rows])
    A, B = oracle.intt(ah, T, p, nthreads=4), oracle.intt(bh, T, p, nthreads=4)
    want = oracle.ntt(oracle.pointwise(A, B, p, n % p), T, p, nthreads=4)
    c = pl.polymul_negacyclic(a.clone(), b.clone())
    assert np.array_equal(eng.to_host(c[rows]), want)
    if logn <= 8:
        assert np.array_equal(want[:3], np.stack([oracle.negacyclic_schoolbook(ah[i], bh[i], p) for i in range(3)]).astype(np.uint64))
    assert pl.count_noncanonical(c) == 0
    one = torch.zeros_like(b)
    one[:, 0] = 1
    assert torch.equal(pl.polymul_negacyclic(a.clone(), one), a)


@pytest.mark.parametrize("p,g", [(998244353, 3), (3221225473, 5)])
@pytest.mark.parametrize("logn,batch", [(6, 300007), (12, 16384), (13, 8209), (16, 2048), (20, 64), (21, 33)])
def test_product_four_byte_words_full_batch(eng, oracle, p, g, logn, batch):
    """The product kernel for 4-byte words (kernels_m32_product.hip): one launch for N <= 2^12 (several polynomials per
    workgroup, ragged tail), fused middle + column passes above; lazy and carry-select butterfly streams."""
    import torch

    n = 1 << logn
    pl = eng.NTTPlan(logn, p, 4, 0)
    T = pl.make_table(2, g)
    pl.set_twiddles(T)
    a = _device_batch(torch, batch, n, p, 4, logn)
    b = _device_batch(torch, batch, n, p, 4, logn + 7)
    rows = [0, 1, batch // 2, batch - 2, batch - 1]
    ah, bh = eng.to_host(a[rows]), eng.to_host(b[rows])
    A, B = oracle.intt(ah, T, p, nthreads=4), oracle.intt(bh, T, p, nthreads=4)
    want = oracle.ntt(oracle.pointwise(A, B, p, n % p), T, p, nthreads=4)
    c = pl.polymul_negacyclic(a.clone(), b.clone())
    assert np.array_equal(eng.to_host(c[rows]), want)
    assert pl.count_noncanonical(c) == 0
    one = torch.zeros_like(b)
    one[:, 0] = 1
    assert torch.equal(pl.polymul_negacyclic(a.clone(), one), a)


def test_column_pass_beyond_grid_limit_is_sliced(eng, oracle):
    """With at most 4 polynomials per column-pass workgroup, N = 2^13 at batch 263140 needs more than 65535 rows of
    workgroups: the launcher slices the batch (pass_kernel.inc: launch_cfg).  17 GB per buffer; sampled rows on both sides
    of every slice boundary against the oracle, the whole batch by round trip."""
    import torch

    p, logn = GOLD, 13
    n = 1 << logn
    batch = 65535 * 4 + 1000
    pl = eng.NTTPlan(logn, p, 8, 0)
    T = pl.make_roots(7)
    pl.set_twiddles(T)
    x = _device_batch(torch, batch, n, p, 8, 13)
    X = pl.forward(x)
    rows = [0, 1, 65535 * 4 - 1, 65535 * 4, 65535 * 4 + 1, batch - 1]
    assert np.array_equal(eng.to_host(X[rows]), oracle.ntt(eng.to_host(x[rows]), T, p, nthreads=4))
    back = pl.inverse(X)
    assert torch.equal(back, x)


@pytest.mark.parametrize("wb,p,g,logn,batch", [(8, GOLD, 7, 16, 4099), (8, GOLD, 7, 13, 33001), (8, GOLD, 7, 12, 16391), (8, GOLD, 7, 18, 1027),
                                               (4, 3221225473, 5, 12, 65539), (4, 998244353, 3, 16, 8197), (4, 3329, 3, 8, 1048583),
                                               (4, 12289, 11, 5, 3000017), (4, 3329, 3, 4, 5000011), (8, GOLD, 7, 3, 3000017),
                                               (8, GOLD, 7, 1, 7000003), (4, 3329, 3, 2, 9000011)])
def test_tapered_launch_ragged_batches(eng, oracle, wb, p, g, logn, batch):
    """Long launches end with rows that stream ppw/2, ppw/4, ppw/8 polynomial groups (pass.h: struct Taper), and the linear tile
    copies rely on the range check of a buffer descriptor that ends at the end of the [batch][N] buffer to drop the chunks of
    polynomials past a ragged batch.  Odd (prime) batches at sizes of every kernel family: rows on both sides of every taper
    level's first group against the oracle, the whole batch by round trip, and a guard polynomial behind the batch -- in the
    output of the forward, the inverse and the in-place transform -- must come back untouched."""
    import torch

    n = 1 << logn
    pl = eng.NTTPlan(logn, p, wb, 0)
    T = pl.make_roots(g)
    pl.set_twiddles(T)
    x = _device_batch(torch, batch + 1, n, p, wb, logn + batch % 97)  # one guard polynomial behind the batch
    guard = x[batch].clone()
    out = torch.full_like(x, 0x5A5A5A5A)
    X = pl.forward(x[:batch], out[:batch])
    assert X.data_ptr() == out.data_ptr()
    marks = sorted({0, 1, batch - 1, batch - 2, *(int(batch * f) + d for f in (0.5, 0.75, 0.875, 0.9375) for d in (-1, 0, 1))})
    assert np.array_equal(eng.to_host(X[marks]), oracle.ntt(eng.to_host(x[marks]), T, p, nthreads=8))
    assert bool((out[batch] == 0x5A5A5A5A).all()), "forward wrote behind the batch"
    back = torch.full_like(x, 0x3C3C3C3C)
    pl.inverse(X, back[:batch])
    assert torch.equal(back[:batch], x[:batch])
    assert bool((back[batch] == 0x3C3C3C3C).all()), "inverse wrote behind the batch"
    pl.forward(x[:batch], x[:batch])  # in place
    assert torch.equal(x[:batch], X) and torch.equal(x[batch], guard)


@pytest.mark.parametrize("wb,p,g,logn,batch", [(4, 3329, 3, 4, 2000003), (4, 3221225473, 5, 5, 1000003), (4, 998244353, 3, 6, 500009),
                                               (8, GOLD, 7, 4, 1000003), (8, GOLD, 7, 5, 500009), (8, GOLD, 7, 7, 100003)])
def test_small_units_block16_layout_full_batch(eng, oracle, wb, p, g, logn, batch):
    """Units whose direct accesses would move less than one 128-byte line per polynomial are staged through LDS in both
    directions (PassCfg::LINEAR_BOTH); there the AIE_BLOCK16 order (src/test.cpp:69-71) is applied inside the tile
    (phase_lds_write / phase_lds_read with perm) instead of by the global store / load: forward to block order and the inverse
    from it, sampled rows against the oracle and the whole odd batch by round trip."""
    import torch

    n = 1 << logn
    pl = eng.NTTPlan(logn, p, wb, 0)
    T = pl.make_roots(g)
    pl.set_twiddles(T)
    x = _device_batch(torch, batch, n, p, wb, 31 * logn + wb)
    rows = [0, 1, 255, 256, batch // 2, batch - 257, batch - 2, batch - 1]
    X = pl.forward(x, layout=eng.LAYOUT_AIE_BLOCK16)
    assert np.array_equal(eng.to_host(X[rows]), oracle.block16(oracle.ntt(eng.to_host(x[rows]), T, p, nthreads=4)))
    assert torch.equal(pl.inverse(X, layout=eng.LAYOUT_AIE_BLOCK16), x)
    Xn = pl.forward(x)
    assert np.array_equal(eng.to_host(Xn[rows]), oracle.ntt(eng.to_host(x[rows]), T, p, nthreads=4))
