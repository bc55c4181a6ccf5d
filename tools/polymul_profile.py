#!/usr/bin/env python3
"""The negacyclic product of BASELINE config 4 (N = 2^20, Goldilocks, batch 512), looped: the program `rocprofv3 --kernel-trace --stats`
wraps to get the per-kernel split of ntt_polymul_negacyclic.  usage: polymul_profile.py [logn] [batch] [reps] [p g]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ntt_aie_amd import NTTPlan

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 512
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
p, g = (int(sys.argv[4], 0), int(sys.argv[5])) if len(sys.argv) > 5 else (0xFFFFFFFF00000001, 7)
plan = NTTPlan(logn, p, 8, 0)
plan.generate_twiddles(2, g)
gen = torch.Generator(device="cuda:0").manual_seed(1)
hi = min(1 << 62, p)
a = torch.randint(0, hi, (batch, 1 << logn), dtype=torch.int64, device="cuda:0", generator=gen)
b = torch.randint(0, hi, (batch, 1 << logn), dtype=torch.int64, device="cuda:0", generator=gen)
c = torch.empty_like(a)
for _ in range(3):
    plan.polymul_negacyclic(a, b, c)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    plan.polymul_negacyclic(a, b, c)
torch.cuda.synchronize()
print("polymul p=%#x N=2^%d batch %d: %.3f ms per product batch" % (p, logn, batch, (time.perf_counter() - t0) / reps * 1e3))
