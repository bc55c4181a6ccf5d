#!/usr/bin/env python3
"""Throughput of every BASELINE.json configuration that fits one GPU (not the driver's bench.py:
that one reports the headline config only).  Prints one JSON object per config."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _explib
_explib.select(default=None)  # the product library unless NTT_HIP_LIB names another build
import numpy as np, torch
from ntt_aie_amd import NTTPlan

GOLD = 0xFFFFFFFF00000001

def rand(batch, n, wb, p, seed):
    g = torch.Generator(device="cuda:0").manual_seed(seed)
    if wb == 8:
        hi = torch.randint(0, min(0xFFFFFFFF, max(1, p >> 32)), (batch, n), dtype=torch.int64, device="cuda:0", generator=g)  # canonical: < p
        lo = torch.randint(0, 1 << 32, (batch, n), dtype=torch.int64, device="cuda:0", generator=g)
        return (hi << 32) | lo
    return torch.randint(0, p, (batch, n), dtype=torch.int64, device="cuda:0", generator=g).to(torch.int32)

def timeit(fn, steps=20, warmup=8):
    for _ in range(warmup): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps

def run(name, logn, p, g, wb, batch, kind=0, polymul=False):
    n = 1 << logn
    plan = NTTPlan(logn, p, wb, 0)
    plan.set_twiddles(plan.make_table(kind, g))
    x = rand(batch, n, wb, p, 1); y = torch.empty_like(x)
    tf = timeit(lambda: plan.forward(x, y))
    ti = timeit(lambda: plan.inverse(x, y))
    out = {"config": name, "logn": logn, "word_bytes": wb, "batch": batch, "hbm_passes": plan.hbm_passes,
           "forward_ms": tf * 1e3, "forward_NTT_per_s": batch / tf, "forward_butterflies_per_s": batch / tf * (n // 2) * logn,
           "forward_alg_GBs": 2 * n * wb * batch / tf / 1e9, "inverse_ms": ti * 1e3, "inverse_NTT_per_s": batch / ti,
           "pass_ms": plan.forward_profile(x, y)}
    if polymul:
        a, b = rand(batch, n, wb, p, 2), rand(batch, n, wb, p, 3)
        tp = timeit(lambda: plan.polymul_negacyclic(a, b), steps=5, warmup=1)
        # the operands as ONE [2*batch][N] buffer: the library then runs both operand transforms as one launch per pass
        ab = rand(2 * batch, n, wb, p, 4)
        tc = timeit(lambda: plan.polymul_negacyclic(ab[:batch], ab[batch:]), steps=5, warmup=1)
        best = min(tp, tc)
        out.update({"polymul_ms": best * 1e3, "polymul_ms_separate_operands": tp * 1e3, "polymul_ms_contiguous_operands": tc * 1e3,
                    "polymul_per_s": batch / best, "polymul_alg_GBs_9N": 9 * n * wb * batch / best / 1e9,
                    "polymul_frac_of_8TBs_9N": 9 * n * wb * batch / best / 8e12,
                    "note": "9N = unfused algorithmic bytes (SURVEY 8d): fwd(a) 2N + fwd(b) 2N + pointwise 3N + inverse 2N words"})
    print(json.dumps(out), flush=True)

if __name__ == "__main__":
    run("cfg2: N=2^12, 32-bit prime 3221225473, batch 1024", 12, 3221225473, 5, 4, 1024)
    run("cfg2b: N=2^12, p=12289 (literal-oracle window), batch 1024", 12, 12289, 11, 4, 1024)
    run("cfg2c: N=2^12, 32-bit prime, batch 65536 (saturating)", 12, 3221225473, 5, 4, 65536)
    run("cfg2d: N=2^12, p=998244353 (< 2^30: lazy butterflies), batch 65536", 12, 998244353, 3, 4, 65536)
    run("kyber-like: N=2^8, p=3329, batch 2^20", 8, 3329, 3, 4, 1 << 20)
    run("cfg3: N=2^16, Goldilocks, batch 4096, forward+inverse", 16, GOLD, 7, 8, 4096)
    run("cfg4: N=2^20, Goldilocks, batch 512, negacyclic polymul", 20, GOLD, 7, 8, 512, kind=2, polymul=True)
    run("ref: N=2^11, p=3329, batch 1 (the reference's own launch)", 11, 3329, 3, 4, 1)
