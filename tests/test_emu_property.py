"""Randomised sweep of the host index model against the oracle: sizes, ragged batches, both
directions, both layouts, in place or not, forced polynomials-per-workgroup, every field."""
import numpy as np
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import emu_lib

GOLD = 0xFFFFFFFF00000001
FIELDS = [(8, GOLD, 7), (4, 3221225473, 5), (4, 998244353, 3), (4, 3329, 3)]


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(field=st.sampled_from(FIELDS), logn=st.integers(1, 14), batch=st.integers(1, 37), inverse=st.booleans(),
       layout=st.booleans(), inplace=st.booleans(), tw=st.sampled_from([1, 4, 64, 8192]), seed=st.integers(0, 2**31))
def test_index_model_matches_oracle(oracle, field, logn, batch, inverse, layout, inplace, tw, seed):
    wb, p, g = field
    n = 1 << logn
    layout = int(layout and logn >= 4)
    dt = np.uint32 if wb == 4 else np.uint64
    T = oracle.make_roots(n, p, g, wb)
    rng = np.random.default_rng(seed)
    a = (rng.integers(0, 2**63, size=(batch, n), dtype=np.uint64) % np.uint64(p)).astype(dt)
    fwd = oracle.ntt(a, T, p, nthreads=2)
    if not inverse:
        src, want = a.copy(), (oracle.block16(fwd) if layout else fwd)
    else:
        src, want = (oracle.block16(fwd) if layout else fwd).copy(), a
    out = src if inplace else np.zeros_like(a)
    rc = emu_lib.lib().emu_transform(wb, logn, p, T.ctypes.data, src.ctypes.data, out.ctypes.data, batch,
                                     int(inverse), layout, 1, tw, 0)
    assert rc == 0
    assert np.array_equal(out, want)
