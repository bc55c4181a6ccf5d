#!/usr/bin/env python3
"""Experiment: does the relative placement of the input and output buffers matter (HBM channel / bank phase)?
Per-pass time of the headline transform with the output buffer shifted by DELTA bytes inside one allocation."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ntt_aie_amd import NTTPlan

GOLD = 0xFFFFFFFF00000001
logn, batch = 16, 4096
n = 1 << logn
words = batch * n
plan = NTTPlan(logn, GOLD, 8, 0)
plan.generate_twiddles(0, 7)
pool = torch.empty(2 * words + (64 << 20) // 8, dtype=torch.int64, device="cuda:0")
g = torch.Generator(device="cuda:0").manual_seed(1)
pool[:words] = torch.randint(0, 1 << 62, (words,), dtype=torch.int64, device="cuda:0", generator=g)
x = pool[:words].view(batch, n)
for delta in [0, 128, 256, 1024, 4096, 65536, 1 << 20, (1 << 20) + 4096, 33 << 20]:
    y = pool[words + delta // 8: 2 * words + delta // 8].view(batch, n)
    for _ in range(3):
        plan.forward(x, y)
    runs = [plan.forward_profile(x, y) for _ in range(7)]
    best = min(runs, key=sum)
    print(json.dumps({"delta_bytes": delta, "pass_ms": [round(m, 4) for m in best], "total_ms": round(sum(best), 4),
                      "median_total": round(sorted(sum(r) for r in runs)[3], 4)}), flush=True)
# in place
for _ in range(3):
    plan.forward(x, x)
runs = [plan.forward_profile(x, x) for _ in range(7)]
best = min(runs, key=sum)
print(json.dumps({"in_place": True, "pass_ms": [round(m, 4) for m in best], "total_ms": round(sum(best), 4)}), flush=True)
