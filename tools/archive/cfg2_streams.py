#!/usr/bin/env python3
"""Experiment for BASELINE config 2 (N = 2^12, 32-bit prime, batch 1024: one generation of workgroups in lockstep): the batch as
K slices on K streams inside one hipGraph, so that the slices' load / compute / store phases are staggered by the launch gaps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ntt_aie_amd import NTTPlan

def timeit(fn, steps=200, warmup=30):
    for _ in range(warmup): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e6

for p, g in ((3221225473, 5), (12289, 11)):
    plan = NTTPlan(12, p, 4, 0); plan.generate_twiddles(0, g)
    for batch in (1024, 2048):
        x = torch.randint(0, p, (batch, 4096), dtype=torch.int64, device="cuda:0").to(torch.int32)
        y = torch.empty_like(x)
        ref = plan.forward(x)
        REP = 20
        for k in (1, 2, 4):
            streams = [torch.cuda.Stream() for _ in range(k)]
            g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1):
                main = torch.cuda.current_stream()
                for _ in range(REP):
                    for s in streams: s.wait_stream(main)
                    sl = batch // k
                    for i, s in enumerate(streams):
                        with torch.cuda.stream(s):
                            plan.forward(x[i * sl:(i + 1) * sl], y[i * sl:(i + 1) * sl], stream=s)
                    for s in streams: main.wait_stream(s)
            us = timeit(g1.replay, steps=20, warmup=3) / REP
            print("p=%d batch=%d slices/streams=%d: %.2f us per transform-of-the-batch  %s" % (p, batch, k, us, "ok" if torch.equal(y, ref) else "MISMATCH"), flush=True)
