#!/usr/bin/env python3
"""Throughput against the batch at one size (looks for bad regimes between one generation of workgroups and saturation):
forward, inverse, forward to AIE_BLOCK16, forward in place.  usage: batch_sweep.py word_bytes logn [max_log_batch]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench_configs as B
from ntt_aie_amd import NTTPlan, LAYOUT_AIE_BLOCK16

wb, logn = int(sys.argv[1]), int(sys.argv[2])
p, g = (B.GOLD, 7) if wb == 8 else (3221225473, 5)
n = 1 << logn
maxlb = int(sys.argv[3]) if len(sys.argv) > 3 else max(0, 31 - logn - (3 if wb == 8 else 2))
plan = NTTPlan(logn, p, wb, 0)
plan.set_twiddles(plan.make_table(0, g))
for lb in range(0, maxlb + 1):
    for batch in sorted({1 << lb, (1 << lb) + (1 << lb >> 1)} if lb else {1}):
        x = B.rand(batch, n, wb, p, 1)
        y = torch.empty_like(x)
        steps = 200 if batch * n < (1 << 22) else 20
        tf = B.timeit(lambda: plan.forward(x, y), steps=steps, warmup=5)
        ti = B.timeit(lambda: plan.inverse(x, y), steps=steps, warmup=5)
        tl = B.timeit(lambda: plan.forward(x, y, layout=LAYOUT_AIE_BLOCK16), steps=steps, warmup=5) if logn >= 4 else 0.0
        tp = B.timeit(lambda: plan.forward(y, y), steps=steps, warmup=5)
        print(json.dumps({"wb": wb, "logn": logn, "batch": batch, "fwd_us": round(tf * 1e6, 2), "inv_us": round(ti * 1e6, 2), "fwd_block16_us": round(tl * 1e6, 2),
                          "fwd_inplace_us": round(tp * 1e6, 2), "fwd_ns_per_poly": round(tf * 1e9 / batch, 1), "fwd_TBs_alg": round(2 * n * wb * batch / tf / 1e12, 3)}), flush=True)
        del x, y
