"""CPU-side checks of the host helpers that mirror the reference's procedure and data formats
(ntt_aie_amd/host.py) and of the pass geometry the launcher derives."""
import numpy as np
import pytest


def test_block_order_matches_reference_rule(oracle):
    from ntt_aie_amd import host

    a = np.arange(2048, dtype=np.uint32)
    assert np.array_equal(host.block_order(a), oracle.block16(a))
    if oracle.have_ref():  # the literal src/test.cpp:212-219 loop
        assert np.array_equal(host.block_order(a).astype(np.int32), oracle.ref_block_order(a.astype(np.int32)))
    b = np.arange(2 * 64, dtype=np.uint64).reshape(2, 64)
    assert np.array_equal(host.block_order(b), oracle.block16(b))
    assert host.ANS_ORDER == oracle.ANS_ORDER


def test_profile_formats():
    from ntt_aie_amd import host

    # profile/plot_exectime.py:27-29 drops every sample equal to the max or the min
    assert host.trimmed_mean([1996, 317, 293, 300, 310]) == pytest.approx((317 + 300 + 310) / 3)
    assert host.trimmed_mean([5, 5, 5]) == 5
    assert host.kerneltime_row(2048, 14.3748) == "2048 , 14.37480"   # profile/kerneltime/aie.csv
    # profile/plot_efficiency.py:44-46 with the AIE's 88 GOPS and its N=2048 kernel time
    assert host.efficiency(2048, 14.3748, 88.0) == pytest.approx(0.0979, abs=5e-4)


def test_bench_input_generator_is_splitmix64_mod_p():
    """bench.py's synthetic inputs: a[b][i] = splitmix64(0x9E3779B97F4A7C15 + b*N + i) mod p, checked
    word for word against plain Python integers (torch on the CPU here; the same tensor code runs on the device)."""
    import torch

    import bench

    M, p = (1 << 64) - 1, bench.GOLDILOCKS

    def sm64(x):
        z = (x + 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)

    n, batch, first = 64, 5, 3
    got = bench.synth_batch(torch, batch, n, "cpu", first_row=first).numpy().view(np.uint64)
    want = np.array([[sm64(bench.SEED + (first + b) * n + i) % p for i in range(n)] for b in range(batch)], dtype=np.uint64)
    assert np.array_equal(got, want)
    # the reduction branch: words in [p, 2^64) must come out as u - p
    z = torch.tensor([bench._s64(p), bench._s64(p + 5), bench._s64(M), bench._s64(p - 1), 0], dtype=torch.int64)
    r = torch.where((z < 0) & (z >= -(2**32 - 1)), z + (2**32 - 1), z).numpy().view(np.uint64)
    assert [int(v) for v in r] == [0, 5, M - p, p - 1, 0]
    # chunked generation (rows per chunk < batch) is seamless
    big = bench.synth_batch(torch, 3, 1 << 24, "cpu").numpy().view(np.uint64)
    assert int(big[2, 12345]) == sm64(bench.SEED + 2 * (1 << 24) + 12345) % p


def test_kernel_source_hash_is_stable_and_sensitive(tmp_path):
    from ntt_aie_amd import _lib

    h = _lib.kernel_source_hash()
    assert len(h) == 16 and h == _lib.kernel_source_hash()
