#!/usr/bin/env python3
"""BASELINE config 2 (N = 2^12, 32-bit prime, batch 1024) is one generation of workgroups: 19.7 us per launch with the
load, compute and store phases of all 1024 workgroups in lockstep.  Sweep of the knobs that could overlap them
(experiment build): fewer, longer workgroups (NTT_TARGET_WGS -> polynomials streamed per workgroup)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _explib
_explib.select()  # the experiment build unless NTT_HIP_LIB names another one
import torch
from ntt_aie_amd import NTTPlan

def timeit(fn, steps=200, warmup=20):
    for _ in range(warmup): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e6

for p, g in ((3221225473, 5), (12289, 11)):
    for batch in (1024, 2048, 4096):
        x = torch.randint(0, p, (batch, 4096), dtype=torch.int64, device="cuda:0").to(torch.int32)
        y = torch.empty_like(x)
        for wgs in ("8192", "1024", "512", "256", "128"):
            os.environ["NTT_TARGET_WGS"] = wgs
            plan = NTTPlan(12, p, 4, 0)
            plan.generate_twiddles(0, g)
            us = timeit(lambda: plan.forward(x, y))
            g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1):
                for _ in range(20):
                    plan.forward(x, y)
            ug = timeit(g1.replay, steps=20, warmup=3) / 20
            print("p=%d batch=%d target_wgs=%s: %.2f us per launch (eager), %.2f us (20 launches in one graph)" % (p, batch, wgs, us, ug), flush=True)
