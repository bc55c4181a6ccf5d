#!/usr/bin/env python3
"""polymul_ms against the workgroup-count target of the product kernel (experiment build).  usage: polymul_ppw.py logn batch"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _explib
_explib.select()  # the experiment build unless NTT_HIP_LIB names another one
import torch
from ntt_aie_amd import NTTPlan
from bench import GOLDILOCKS, synth_batch
logn, batch = int(sys.argv[1]), int(sys.argv[2])
a = synth_batch(torch, batch, 1 << logn, torch.device("cuda", 0)); b = a.clone(); c = torch.empty_like(a)
for wgs in (2048, 4096, 8192, 16384, 32768, 65536):
    os.environ["NTT_TARGET_WGS"] = str(wgs); os.environ["NTT_TARGET_WGS_COL"] = "16384"
    plan = NTTPlan(logn, GOLDILOCKS, 8, 0); plan.generate_twiddles(2, 7)
    for _ in range(5): plan.polymul_negacyclic(a, b, c)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(15): plan.polymul_negacyclic(a, b, c)
    torch.cuda.synchronize(); print("logn %d batch %d target %d: %.4f ms" % (logn, batch, wgs, (time.perf_counter() - t0) / 15 * 1e3), flush=True)
