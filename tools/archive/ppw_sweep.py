#!/usr/bin/env python3
"""Polynomials streamed per workgroup (ppw) against the batch, for the first (CONTIG) and second (column) pass: per-pass time for
NTT_TARGET_WGS / NTT_TARGET_WGS_COL targets that give ppw = 2 .. 32 (experiment build).  usage: ppw_sweep.py logn batch[,batch...]"""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _explib
_explib.select()  # the experiment build unless NTT_HIP_LIB names another one
import torch
from ntt_aie_amd import NTTPlan
from bench import GOLDILOCKS, synth_batch

logn = int(sys.argv[1])
for batch in [int(b) for b in sys.argv[2].split(",")]:
    x = synth_batch(torch, batch, 1 << logn, torch.device("cuda", 0)); y = torch.empty_like(x)
    rows = []
    for wgs in (2048, 4096, 8192, 16384, 32768, 65536):
        os.environ["NTT_TARGET_WGS"] = str(wgs); os.environ["NTT_TARGET_WGS_COL"] = str(wgs)
        plan = NTTPlan(logn, GOLDILOCKS, 8, 0); plan.generate_twiddles(0, 7)
        for _ in range(5): plan.forward(x, y)
        s = [plan.forward_profile(x, y) for _ in range(21)]
        rows.append("%d:%s" % (wgs, "+".join("%.4f" % statistics.median(v[i] for v in s) for i in range(len(s[0])))))
    print("logn=%d batch=%d ms per pass by target: %s" % (logn, batch, "  ".join(rows)), flush=True)
    del x, y
