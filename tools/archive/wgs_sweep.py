#!/usr/bin/env python3
"""Sweep of the workgroups-per-launch target (experiment build, NTT_TARGET_WGS / NTT_TARGET_WGS_COL) against the batch:
per-pass kernel time (hipEvents, median of 15).  usage: wgs_sweep.py wb logn batch[,batch...]"""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _explib
_explib.select()  # the experiment build unless NTT_HIP_LIB names another one
import torch
from ntt_aie_amd import NTTPlan

wb, logn = int(sys.argv[1]), int(sys.argv[2])
p, g = (0xFFFFFFFF00000001, 7) if wb == 8 else (3221225473, 5)
for batch in [int(b) for b in sys.argv[3].split(",")]:
    x = torch.randint(0, p if wb == 4 else 1 << 62, (batch, 1 << logn), dtype=torch.int64, device="cuda:0")
    if wb == 4:
        x = x.to(torch.int32)
    y = torch.empty_like(x)
    row = []
    for wgs in (512, 1024, 2048, 4096, 8192, 16384):
        os.environ["NTT_TARGET_WGS"] = str(wgs)
        os.environ["NTT_TARGET_WGS_COL"] = str(wgs)
        plan = NTTPlan(logn, p, wb, 0)
        plan.generate_twiddles(0, g)
        for _ in range(5):
            plan.forward(x, y)
        s = [plan.forward_profile(x, y) for _ in range(15)]
        med = [statistics.median(v[i] for v in s) * 1e3 for i in range(len(s[0]))]
        row.append("%d:%s" % (wgs, "+".join("%.1f" % m for m in med)))
    print("wb=%d logn=%d batch=%d us per pass by target: %s" % (wb, logn, batch, "  ".join(row)), flush=True)
