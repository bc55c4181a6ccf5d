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
