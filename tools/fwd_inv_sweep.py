#!/usr/bin/env python3
"""Experiment: forward / inverse / product wall-clock per call at 4 GiB of Goldilocks coefficients (or 4-byte words
with wb=4).  usage: fwd_inv_sweep.py wb logn [logn ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _explib
_explib.select(default=None)  # the product library unless NTT_HIP_LIB names another build
import torch
from ntt_aie_amd import NTTPlan

wb = int(sys.argv[1])
p, g = (0xFFFFFFFF00000001, 7) if wb == 8 else (int(os.environ.get("NTT_SWEEP_P", "3221225473")), int(os.environ.get("NTT_SWEEP_G", "5")))


def timeit(fn, steps=8, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for logn in map(int, sys.argv[2:]):
    batch = (1 << (32 - (3 if wb == 8 else 2))) >> logn
    gen = torch.Generator(device="cuda:0").manual_seed(1)
    x = torch.randint(0, p if wb == 4 else 1 << 62, (batch, 1 << logn), dtype=torch.int64, device="cuda:0", generator=gen)
    if wb == 4:
        x = x.to(torch.int32)
    y = torch.empty_like(x)
    plan = NTTPlan(logn, p, wb, 0)
    plan.generate_twiddles(1, g)
    out = {"lib": os.path.basename(os.environ.get("NTT_HIP_LIB", "default")), "wb": wb, "logn": logn, "batch": batch,
           "fwd_ms": round(timeit(lambda: plan.forward(x, y)), 3), "inv_ms": round(timeit(lambda: plan.inverse(x, y)), 3)}
    del x, y
    print(json.dumps(out), flush=True)
