#!/usr/bin/env python3
"""Throughput of every BASELINE.json configuration that fits one GPU (not the driver's bench.py: that one reports the
headline config only), one JSON object per line, each with a `roofline` object carrying the same keys as bench.py's
(`bound`, `achieved`, `peak`, `unit`, `frac`, `frac_ceiling`, `traffic`, `valu`) -- made by the same function, bench.config_roofline.

Timing is live (hipEvents on the launch stream around a loop of operations, and ntt_forward_profile per pass).  Counter
figures (`traffic`, `valu`) are quoted from profiles/<round>_<cfg>_pmc_traffic.json / _sq_counters.json -- written by
tools/collect_profiles.sh -- and only when their kernel-source hash equals this tree's (null + reason otherwise)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _explib

_explib.select(default=None)  # the product library unless NTT_HIP_LIB names another build
import torch

import bench
from configs import CONFIGS, GOLD, algorithmic_bytes, butterflies
from ntt_aie_amd import NTTPlan, _lib



def rand(batch, n, wb, p, seed):
    g = torch.Generator(device="cuda:0").manual_seed(seed)
    if wb == 8:
        hi = torch.randint(0, min(0xFFFFFFFF, max(1, p >> 32)), (batch, n), dtype=torch.int64, device="cuda:0", generator=g)  # canonical: < p
        lo = torch.randint(0, 1 << 32, (batch, n), dtype=torch.int64, device="cuda:0", generator=g)
        return (hi << 32) | lo
    return torch.randint(0, p, (batch, n), dtype=torch.int64, device="cuda:0", generator=g).to(torch.int32)


def timeit(fn, steps=20, warmup=8):
    """ms per call: hipEvents around `steps` back-to-back calls on the current stream (resident buffers, no host sync inside)."""
    for _ in range(warmup):
        fn()
    s = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(steps):
        fn()
    e1.record(s)
    e1.synchronize()
    return e0.elapsed_time(e1) / steps


def roofline(cfg_key, c, op_ms, pass_ms, copy_ms, passes):
    """bench.py's roofline object for one configuration (bench.config_roofline: the same function the driver-run line uses).
    A transform is priced on the summed hipEvent durations of its pass kernels when ntt_forward_profile gave them."""
    t = sum(pass_ms) if pass_ms else op_ms
    return bench.config_roofline(cfg_key, c, t, copy_ms, passes, _lib.kernel_source_hash())


def run(name, logn, p, g, wb, batch, kind=0, polymul=False, cfg_key=None):
    n = 1 << logn
    plan = NTTPlan(logn, p, wb, 0)
    plan.set_twiddles(plan.make_table(kind, g))
    x = rand(batch, n, wb, p, 1)
    y = torch.empty_like(x)
    tf = timeit(lambda: plan.forward(x, y))
    ti = timeit(lambda: plan.inverse(x, y))
    pass_ms = [0.0] * plan.hbm_passes
    for _ in range(5):
        plan.forward(x, y)
        plan.forward(x, y)
        pass_ms = [a + b / 5 for a, b in zip(pass_ms, plan.forward_profile(x, y))]
    out = {"config": name, "logn": logn, "word_bytes": wb, "batch": batch, "hbm_passes": plan.hbm_passes,
           "forward_ms": tf, "forward_NTT_per_s": batch / (tf * 1e-3), "forward_butterflies_per_s": batch / (tf * 1e-3) * (n // 2) * logn,
           "forward_alg_GBs": 2 * n * wb * batch / (tf * 1e-3) / 1e9, "inverse_ms": ti, "inverse_NTT_per_s": batch / (ti * 1e-3),
           "pass_ms": pass_ms}
    op_ms = tf
    if polymul:
        a, b = rand(batch, n, wb, p, 2), rand(batch, n, wb, p, 3)
        tp = timeit(lambda: plan.polymul_negacyclic(a, b), steps=5, warmup=2)
        # the operands as ONE [2*batch][N] buffer: the library then runs both operand transforms as one launch per pass
        ab = rand(2 * batch, n, wb, p, 4)
        tc = timeit(lambda: plan.polymul_negacyclic(ab[:batch], ab[batch:]), steps=5, warmup=2)
        best = min(tp, tc)
        out.update({"polymul_ms": best, "polymul_ms_separate_operands": tp, "polymul_ms_contiguous_operands": tc,
                    "polymul_per_s": batch / (best * 1e-3), "polymul_alg_GBs_9N": 9 * n * wb * batch / (best * 1e-3) / 1e9,
                    "polymul_frac_of_8TBs_9N": 9 * n * wb * batch / (best * 1e-3) / 8e12,
                    "note": "9N = unfused algorithmic bytes (SURVEY 8d): fwd(a) 2N + fwd(b) 2N + pointwise 3N + inverse 2N words"})
        op_ms = best
        del a, b, ab
    if cfg_key:
        c = CONFIGS[cfg_key]
        assert (c["logn"], c["p"], c["wb"], c["batch"]) == (logn, p, wb, batch)
        # a device copy of the operation's algorithmic bytes, same process: what plain streaming of that much data achieves here
        words = algorithmic_bytes(c) // 2 // wb
        src = torch.empty(words, dtype=x.dtype, device="cuda:0")
        dst = torch.empty_like(src)
        copy_ms = timeit(lambda: dst.copy_(src), steps=10, warmup=3)
        del src, dst
        out["roofline"] = roofline(cfg_key, c, op_ms, None if polymul else pass_ms, copy_ms, plan.hbm_passes)
        out["butterflies_per_s"] = butterflies(c) / (op_ms * 1e-3)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    run("cfg2: N=2^12, 32-bit prime 3221225473, batch 1024", 12, 3221225473, 5, 4, 1024, cfg_key="cfg2")
    run("cfg2b: N=2^12, p=12289 (literal-oracle window), batch 1024", 12, 12289, 11, 4, 1024)
    run("cfg2c: N=2^12, 32-bit prime, batch 65536 (saturating)", 12, 3221225473, 5, 4, 65536, cfg_key="cfg2_sat")
    run("cfg2d: N=2^12, p=998244353 (< 2^30: lazy butterflies), batch 65536", 12, 998244353, 3, 4, 65536)
    run("kyber-like: N=2^8, p=3329, batch 2^20", 8, 3329, 3, 4, 1 << 20)
    run("cfg3: N=2^16, Goldilocks, batch 4096, forward+inverse (roofline object: bench.py's line)", 16, GOLD, 7, 8, 4096)
    run("cfg4: N=2^20, Goldilocks, batch 512, negacyclic polymul", 20, GOLD, 7, 8, 512, kind=2, polymul=True, cfg_key="cfg4")
    run("ref: N=2^11, p=3329, batch 1 (the reference's own launch)", 11, 3329, 3, 4, 1)
