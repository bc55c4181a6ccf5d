#!/usr/bin/env python3
"""Determinism soak of the kernels whose exchanges mix wave-local syncs with workgroup barriers (12- and 13-stage passes, the
product pass): the same launch repeated many times must give the same words every time (a missing barrier shows up as an
occasional difference, long before it shows up in a parity test), and the first result is checked by the round trip.
usage: race_soak.py [repeats=150]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench_configs as B
from ntt_aie_amd import NTTPlan

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
shapes = [(4, 3221225473, 5, 12, 4099), (4, 998244353, 3, 13, 2053), (4, 3221225473, 5, 21, 5), (8, B.GOLD, 7, 12, 2053), (8, B.GOLD, 7, 20, 37),
          (8, B.GOLD, 7, 21, 9), (8, B.GOLD, 7, 18, 131), (4, 998244353, 3, 20, 67), (8, B.GOLD, 7, 16, 1031), (4, 12289, 11, 8, 100003),
          # round 3: the 13-stage alternative of 8-byte N = 2^13 and the 14-stage one of lazy 4-byte N = 2^14 (batches above their thresholds),
          # the 9-stage column pass (N = 2^22, three fields), the general 64-bit modulus (generated streams, folded scaling)
          (8, B.GOLD, 7, 13, 1031), (4, 998244353, 3, 14, 1031), (8, B.GOLD, 7, 22, 5), (4, 998244353, 3, 22, 9), (4, 3221225473, 5, 22, 2),
          (8, 0x3FFFFFEE00000001, 3, 16, 517), (8, 0xFFFFFFFC00000001, 10, 12, 2053), (8, 0x3FFFFFEE00000001, 3, 21, 5),
          # round 4: the 512-thread x 8-word variant of the single-pass sizes (batches below its threshold: four exchanges, the last
          # one across waves at 2^12), every 4-byte modulus class, Goldilocks (LDS-DMA forward as the only pass) and the general modulus
          (4, 3221225473, 5, 12, 1031), (4, 2013265921, 31, 11, 517), (4, 998244353, 3, 12, 257), (4, 3221225473, 5, 10, 2047),
          (8, B.GOLD, 7, 12, 1031), (8, B.GOLD, 7, 10, 2047), (8, 0x3FFFFFEE00000001, 3, 12, 517)]
bad = 0
for wb, p, g, logn, batch in shapes:
    n = 1 << logn
    plan = NTTPlan(logn, p, wb, 0)
    plan.set_twiddles(plan.make_table(2, g))
    x, b = B.rand(batch, n, wb, p, 1), B.rand(batch, n, wb, p, 2)
    f0 = plan.forward(x)
    assert torch.equal(plan.inverse(f0), x), "round trip"
    i0 = plan.inverse(x)
    c0 = plan.polymul_negacyclic(x.clone(), b.clone())
    diff = 0
    y = torch.empty_like(x)
    for r in range(reps):
        diff += int(not torch.equal(plan.forward(x, y), f0))
        diff += int(not torch.equal(plan.inverse(x, y), i0))
        if r % 5 == 0:
            diff += int(not torch.equal(plan.polymul_negacyclic(x.clone(), b.clone()), c0))
    print("wb=%d p=%d logn=%d batch=%d passes=%s: %d differing results in %d repeats" % (wb, p, logn, batch, plan.passes_for(batch), diff, reps), flush=True)
    bad += diff
    del x, b, y, f0, i0, c0
sys.exit(1 if bad else 0)
