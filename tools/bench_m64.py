#!/usr/bin/env python3
"""Throughput of the general odd 64-bit modulus (FieldM64, Montgomery R = 2^64) beside Goldilocks at the headline shape
(N = 2^16, batch 4096) and at N = 2^12 / 2^20: forward and inverse ms per call, NTT/s, butterflies/s."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ntt_aie_amd import NTTPlan

FIELDS = [("goldilocks 2^64-2^32+1", 0xFFFFFFFF00000001, 7), ("62-bit 0x3fffffee00000001", 0x3FFFFFEE00000001, 3),
          ("64-bit 0xfffffffc00000001", 0xFFFFFFFC00000001, 10)]


def timeit(fn, steps=20, warmup=25):  # a long warm-up: the first field measured would otherwise pay the clock ramp
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for logn, batch in ((16, 4096), (12, 65536), (20, 256)):
    n = 1 << logn
    gen = torch.Generator(device="cuda:0").manual_seed(1)
    x = torch.randint(0, 1 << 61, (batch, n), dtype=torch.int64, device="cuda:0", generator=gen)  # < every modulus here
    y = torch.empty_like(x)
    for name, p, g in FIELDS:
        plan = NTTPlan(logn, p, 8, 0)
        plan.generate_twiddles(0, g)
        f, i = timeit(lambda: plan.forward(x, y)), timeit(lambda: plan.inverse(x, y))
        print(json.dumps({"field": name, "logn": logn, "batch": batch, "fwd_ms": round(f, 4), "inv_ms": round(i, 4),
                          "fwd_NTT_per_s": round(batch / f * 1e3), "fwd_butterflies_per_s": batch / f * 1e3 * (n // 2) * logn}), flush=True)
