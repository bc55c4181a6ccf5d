#!/usr/bin/env python3
"""30 forward launches of one size at batch 1 (p = 3329, the reference's parameter set) -- the program rocprofv3 wraps
in tools/kerneltime_rocprof.sh to read true kernel durations.  usage: one_size.py LOGN"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ntt_aie_amd import NTTPlan, to_device
logn = int(sys.argv[1]); n = 1 << logn
plan = NTTPlan(logn, 3329, 4, 0); plan.set_twiddles(plan.make_roots(3))
x = to_device((np.arange(n, dtype=np.uint64) % 3329).astype(np.uint32)[None, :], "cuda:0"); y = torch.empty_like(x)
for _ in range(30):
    plan.forward(x, y)
torch.cuda.synchronize()
