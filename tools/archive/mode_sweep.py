#!/usr/bin/env python3
"""Every transform mode (layouts, scaled / unscaled inverse, in place) against N at 1 GiB of coefficients: looks for a mode that falls off the
curve of the plain forward transform."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench_configs as B
from ntt_aie_amd import NTTPlan, LAYOUT_AIE_BLOCK16
for wb, p, g in ((8, B.GOLD, 7), (4, 3221225473, 5)):
    for logn in (4, 6, 8, 10, 12, 13, 16, 20, 21, 22):
        n = 1 << logn
        batch = max(1, (1 << 30) // (n * wb))
        plan = NTTPlan(logn, p, wb, 0)
        plan.generate_twiddles(0, g)
        x = B.rand(batch, n, wb, p, 1); y = torch.empty_like(x)
        r = {"wb": wb, "logn": logn, "batch": batch}
        for name, fn in (("fwd", lambda: plan.forward(x, y)), ("fwd_b16", lambda: plan.forward(x, y, layout=LAYOUT_AIE_BLOCK16)),
                         ("inv", lambda: plan.inverse(x, y)), ("inv_b16", lambda: plan.inverse(x, y, layout=LAYOUT_AIE_BLOCK16)),
                         ("inv_unscaled", lambda: plan.inverse(x, y, scale=False)), ("fwd_inplace", lambda: plan.forward(y, y)),
                         ("inv_inplace", lambda: plan.inverse(y, y))):
            r[name] = round(B.timeit(fn, steps=8, warmup=3) * 1e3, 3)
        print(json.dumps(r), flush=True)
        del x, y
