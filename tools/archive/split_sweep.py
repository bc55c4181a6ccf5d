#!/usr/bin/env python3
"""Experiment: per-pass time of alternative stage splits (NTT_PLAN_SPLIT) at a fixed 4 GiB of coefficients
(Goldilocks; NTT_SWEEP_WB=4 for 4-byte words).  usage: split_sweep.py logn split [split ...]
e.g.  split_sweep.py 20 12,8 8,6,6 7,7,6"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
# experiment knobs live only in libntt_hip_exp.so (make -C ntt_aie_amd/csrc exp): the product library reads no env
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _explib
_explib.select()  # the experiment build unless NTT_HIP_LIB names another one
import torch
from ntt_aie_amd import NTTPlan

GOLD = 0xFFFFFFFF00000001
WB = int(os.environ.get("NTT_SWEEP_WB", "8"))
P, G = (GOLD, 7) if WB == 8 else (3221225473, 5)
logn = int(sys.argv[1])
batch = (1 << (32 - (3 if WB == 8 else 2))) >> logn
g = torch.Generator(device="cuda:0").manual_seed(1)
x = torch.randint(0, 1 << 62 if WB == 8 else P, (batch, 1 << logn), dtype=torch.int64, device="cuda:0", generator=g)
if WB == 4:
    x = x.to(torch.int32)
y = torch.empty_like(x)
for split in sys.argv[2:]:
    os.environ["NTT_PLAN_SPLIT"] = split
    plan = NTTPlan(logn, P, WB, 0)
    plan.generate_twiddles(1, G)
    for _ in range(3):
        plan.forward(x, y)
    best = None
    for _ in range(5):
        ms = plan.forward_profile(x, y)
        if best is None or sum(ms) < sum(best):
            best = ms
    print(json.dumps({"wb": WB, "logn": logn, "batch": batch, "split": split, "passes": plan.hbm_passes,
                      "pass_ms": [round(m, 3) for m in best], "total_ms": round(sum(best), 3)}), flush=True)
