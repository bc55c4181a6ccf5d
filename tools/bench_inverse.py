#!/usr/bin/env python3
"""Inverse transform time at the headline shape and at N = 2^20 (A/B with NTT_HIP_LIB=ab/libntt_NAME.so)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _explib
_explib.select(default=None)  # the product library unless NTT_HIP_LIB names another build
import torch
from bench import GOLDILOCKS, synth_batch
from ntt_aie_amd import NTTPlan
torch.cuda.set_device(0)
for logn, batch in ((16, 4096), (16, 8192), (16, 16384), (13, 65536)):
    plan = NTTPlan(logn, GOLDILOCKS, 8, 0); plan.generate_twiddles(0, 7)
    x = synth_batch(torch, batch, 1 << logn, torch.device("cuda", 0)); y = torch.empty_like(x)
    for _ in range(10): plan.inverse(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): plan.inverse(x, y)
    torch.cuda.synchronize(); print("logn %d batch %d inverse %.4f ms" % (logn, batch, (time.perf_counter() - t0) / 30 * 1e3), flush=True)
    del x, y
