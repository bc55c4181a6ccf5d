#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (data only: inputs + expected words).

Two independent sources, both committed as plain integer arrays:

  literal_*.npz   produced by the LITERAL reference lines (src/test.cpp:15-60,
                  :69-71, :212-219 and src/aie_core.cc:11-39) compiled where they
                  lie into oracle/_ref/libntt_ref.so by oracle/build_ref.sh.
                  Valid for p <= 46340.  Needs /root/reference.
  bigint_*.npz    produced by an independent pure-Python big-int restatement of
                  the same network (this file, `net_forward`), for primes outside
                  the literal code's int32 window: 998244353, 3221225473, the
                  Goldilocks prime 2^64-2^32+1, and (round 3, general 64-bit modulus)
                  the 62-bit NTT prime 0x3fffffee00000001 (g = 3) and the 64-bit one
                  0xfffffffc00000001 (g = 10, above 2^63: sums carry out of the word).

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

GOLD = 0xFFFFFFFF00000001
P62 = 0x3FFFFFEE00000001   # 4611685941117976577 = 0x3fffffee * 2^32 + 1, prime, primitive root 3
P64B = 0xFFFFFFFC00000001  # 18446744056529682433 = 0xfffffffc * 2^32 + 1, prime, primitive root 10


def splitmix64(x):
    """SURVEY 8(d) input generator."""
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def rand_poly(n, p, seed):
    return [splitmix64(seed + i) % p for i in range(n)]


def roots_rule(n, p, g):
    """Table rule a2 in Python ints (test.cpp:27-32, :138)."""
    w = pow(g, (p - 1) // n, p)
    T = [1] * n
    for i in range(1, n):
        T[i] = T[i - 1] * w % p
    return T


def net_forward(a, T, p, stages=None):
    """Network rule a1 in Python ints (test.cpp:34-60)."""
    a = list(a)
    n = len(a)
    t, m, s = 1, n, 0
    while m > 1:
        h = m // 2
        for i in range(h):
            r = T[h + i]
            for j in range(i * 2 * t, i * 2 * t + t):
                v0, v1 = a[j], a[j + t]
                a[j] = (v0 + v1) % p
                a[j + t] = ((v0 - v1) % p) * r % p
        t <<= 1
        m >>= 1
        s += 1
        if stages is not None and s == stages:
            break
    return a


def literal():
    import oracle_py as O

    if not O.have_ref():
        print("literal reference not buildable here; skipping literal_*.npz")
        return
    cases = [(16, 3329, 3), (256, 3329, 3), (2048, 3329, 3), (4096, 12289, 11), (8192, 40961, 3)]
    for n, p, g in cases:
        logn = n.bit_length() - 1
        T = O.ref_make_roots(n, p, g)
        a_iota = (np.arange(n) % p).astype(np.int32)
        a_rand = np.array(rand_poly(n, p, 1000 * n), dtype=np.int32)
        out = {"n": n, "p": p, "g": g, "table": T,
               "in_iota": a_iota, "out_iota": O.ref_ntt(a_iota, T, p, logn - 1),
               "in_rand": a_rand, "out_rand": O.ref_ntt(a_rand, T, p, logn - 1)}
        out["out_iota_block16"] = O.ref_block_order(out["out_iota"])
        if n == 256:  # the test_stage hook (test.cpp:55-58, :67): every partial network
            out["partial_iota"] = np.stack([O.ref_ntt(a_iota, T, p, s) for s in range(logn)])
        np.savez_compressed(os.path.join(HERE, "literal_n%d_p%d.npz" % (n, p)), **out)
        print("literal n=%d p=%d out[0..4]=%s" % (n, p, out["out_iota"][:4]))
    # scalar arithmetic twins aie_core.cc:11-39 with the constants of aie2.py:17-19
    import math
    R = O.ref()
    q = 3329
    w = math.ceil(math.log2(q))
    u = math.floor(pow(2, 2 * w) / q)
    rng = np.random.default_rng(7)
    ab = rng.integers(0, q, size=(4096, 2), dtype=np.int64)
    np.savez_compressed(
        os.path.join(HERE, "literal_scalar_q3329.npz"), q=q, w=w, u=u, ab=ab.astype(np.int32),
        modadd=np.array([R.ref_modadd(int(a), int(b), q) for a, b in ab], dtype=np.int32),
        modsub=np.array([R.ref_modsub(int(a), int(b), q) for a, b in ab], dtype=np.int32),
        barrett=np.array([R.ref_barrett_2k(int(a), int(b), q, w, u) for a, b in ab], dtype=np.int32))


def bigint(only_new=False):
    cases = [(64, 998244353, 3, 4), (1024, 998244353, 3, 4), (256, 3221225473, 5, 4),
             (64, GOLD, 7, 8), (1024, GOLD, 7, 8), (4096, GOLD, 7, 8),
             (64, P62, 3, 8), (4096, P62, 3, 8), (1024, P64B, 10, 8)]
    for n, p, g, wb in cases:
        if only_new and os.path.exists(os.path.join(HERE, "bigint_n%d_p%d.npz" % (n, p))):
            continue
        dt = np.uint32 if wb == 4 else np.uint64
        T = roots_rule(n, p, g)
        a = rand_poly(n, p, 77 * n + wb)
        edge = [0, p - 1, 1, p - 2] * (n // 4)  # extreme residues: carries / borrows everywhere
        np.savez_compressed(
            os.path.join(HERE, "bigint_n%d_p%d.npz" % (n, p)), n=n, p=np.uint64(p), g=g,
            table=np.array(T, dtype=dt), in_rand=np.array(a, dtype=dt),
            out_rand=np.array(net_forward(a, T, p), dtype=dt), in_edge=np.array(edge, dtype=dt),
            out_edge=np.array(net_forward(edge, T, p), dtype=dt))
        print("bigint n=%d p=%d ok" % (n, p))


if __name__ == "__main__":
    if "--only-new" in sys.argv:  # add fixtures that do not exist yet, leave the committed ones byte-identical
        bigint(only_new=True)
    else:
        literal()
        bigint()
