#!/usr/bin/env python3
"""Experiment: what would a pass cost if its coefficient traffic were served by the 256 MiB Infinity Cache instead of HBM?
NTT_DEBUG_FLAGS = 8 | (G << 4) (experiment build) confines every workgroup's loads and stores to the first G polynomials
of the batch, so the same instruction stream runs with a footprint of G x 512 KiB per buffer (outputs meaningless).
G = 4096 is the real transform.  Prints per-pass hipEvent times, out of place and in place.
usage: mall_probe.py [logn=16] [batch=4096]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _explib
_explib.select()  # the experiment build unless NTT_HIP_LIB names another one
import torch
from ntt_aie_amd import NTTPlan

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 16
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
p, g = 0xFFFFFFFF00000001, 7
gen = torch.Generator(device="cuda:0").manual_seed(1)
x = torch.randint(0, 1 << 62, (batch, 1 << logn), dtype=torch.int64, device="cuda:0", generator=gen)
y = torch.empty_like(x)
for G in (batch, 1024, 512, 256, 128, 64, 16, batch):
    os.environ["NTT_DEBUG_FLAGS"] = str(8 | (G << 4)) if G != batch else "0"
    plan = NTTPlan(logn, p, 8, 0)
    plan.generate_twiddles(1, g)
    row = {"logn": logn, "batch": batch, "G": G, "footprint_MB_per_buffer": G * (8 << logn) >> 20}
    for name, dst in (("out_of_place", y), ("in_place", x)):
        for _ in range(3):
            plan.forward(x, dst)
        runs = [plan.forward_profile(x, dst) for _ in range(7)]
        best = min(runs, key=sum)
        row[name + "_pass_ms"] = [round(m, 3) for m in best]
    print(json.dumps(row), flush=True)
