#!/usr/bin/env python3
"""Negacyclic product and pointwise product at small sizes (or the sizes given: polymul_small.py logn ...), 1 GiB per operand: the sizes below the product kernel's range
(Goldilocks N < 2^7, 4-byte words N < 2^5) fold only the pointwise leg into the forward pass.  One JSON line per shape."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench_configs as B
from ntt_aie_amd import NTTPlan

for wb, p, g in ((8, B.GOLD, 7), (4, 998244353, 3)):
    for logn in ([int(v) for v in sys.argv[1:]] or [2, 3, 4, 5, 6, 7, 8, 10]):
        n = 1 << logn
        batch = (1 << 30) // (n * wb)
        plan = NTTPlan(logn, p, wb, 0)
        plan.set_twiddles(plan.make_table(2, g))
        a, b = B.rand(batch, n, wb, p, 2), B.rand(batch, n, wb, p, 3)
        c = torch.empty_like(a)
        tp = B.timeit(lambda: plan.polymul_negacyclic(a, b), steps=5, warmup=2)
        tw = B.timeit(lambda: plan.pointwise_mul(a, b, c), steps=5, warmup=2)
        tf = B.timeit(lambda: plan.forward(a, c), steps=5, warmup=2)
        print(json.dumps({"word_bytes": wb, "logn": logn, "batch": batch, "polymul_ms": round(tp * 1e3, 3), "pointwise_ms": round(tw * 1e3, 3),
                          "forward_ms": round(tf * 1e3, 3), "copy_equiv_ms_3N": round(3 * (1 << 30) / 5.7e12 * 1e3, 3)}), flush=True)
        del a, b, c
