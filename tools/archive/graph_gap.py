#!/usr/bin/env python3
"""Experiment: launch gaps of the headline transform -- back-to-back launches vs one hipGraph of 20 steps vs the sum of
the per-pass kernel times (measured: 1.797 / 1.775 / 1.757 ms per step: a graph recovers about 1 %)."""
import os
import sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ntt_aie_amd import NTTPlan
GOLD = 0xFFFFFFFF00000001
plan = NTTPlan(16, GOLD, 8, 0); plan.generate_twiddles(0, 7)
g = torch.Generator(device="cuda:0").manual_seed(1)
x = torch.randint(0, 1 << 62, (4096, 1 << 16), dtype=torch.int64, device="cuda:0", generator=g)
y = torch.empty_like(x)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(10): plan.forward(x, y, stream=s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): plan.forward(x, y, stream=s)
    torch.cuda.synchronize()
    print("launches  ms/step", (time.perf_counter() - t0) / 20 * 1e3)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        for _ in range(20): plan.forward(x, y, stream=s)
    gr.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter(); gr.replay(); torch.cuda.synchronize()
    print("graph(20) ms/step", (time.perf_counter() - t0) / 20 * 1e3)
    print("pass sum  ms     ", sum(plan.forward_profile(x, y, stream=s)))
