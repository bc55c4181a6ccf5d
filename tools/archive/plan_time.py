#!/usr/bin/env python3
"""Time of plan creation, device-side table generation (ntt_plan_generate_twiddles) and the host path (make_table + set_twiddles with
its batch inversion and upload) against N, both word sizes."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ntt_aie_amd import NTTPlan
GOLD = 0xFFFFFFFF00000001
for wb, p, g in ((8, GOLD, 7), (4, 998244353, 3)):
    for logn in (12, 16, 20, 22) + ((24, 26) if wb == 8 else ()):
        t0 = time.perf_counter(); pl = NTTPlan(logn, p, wb, 0); t1 = time.perf_counter()
        pl.generate_twiddles(2 if logn <= (32 if wb == 8 else 22) else 0, g); torch.cuda.synchronize(); t2 = time.perf_counter()
        T = pl.make_table(0, g); t3 = time.perf_counter()
        pl.set_twiddles(T); torch.cuda.synchronize(); t4 = time.perf_counter()
        print("wb=%d logn=%d: create %.1f ms, generate_twiddles (device) %.1f ms, make_table (host) %.1f ms, set_twiddles (host inverse + upload) %.1f ms" % (
            wb, logn, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3), flush=True)
        del pl
