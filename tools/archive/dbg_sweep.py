#!/usr/bin/env python3
"""Experiment: per-pass time under NTT_DEBUG_FLAGS (1: loads hit L2, 2: no stores, 3: both = VALU floor).
usage: dbg_sweep.py logn word_bytes [split]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
# experiment knobs live only in libntt_hip_exp.so (make -C ntt_aie_amd/csrc exp): the product library reads no env
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _explib
_explib.select()  # the experiment build unless NTT_HIP_LIB names another one
import torch
from ntt_aie_amd import NTTPlan

logn, wb = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3:
    os.environ["NTT_PLAN_SPLIT"] = sys.argv[3]
p, g = (0xFFFFFFFF00000001, 7) if wb == 8 else ((3221225473, 5) if os.environ.get("NTT_SWEEP_SMALL_P") != "1" else (12289, 11))
batch = (1 << (32 - (3 if wb == 8 else 2))) >> logn  # 4 GiB of coefficients
gen = torch.Generator(device="cuda:0").manual_seed(1)
x = torch.randint(0, p if wb == 4 else 1 << 62, (batch, 1 << logn), dtype=torch.int64, device="cuda:0", generator=gen)
if wb == 4:
    x = x.to(torch.int32)
y = torch.empty_like(x)
for flags in (0, 1, 2, 3):
    os.environ["NTT_DEBUG_FLAGS"] = str(flags)
    plan = NTTPlan(logn, p, wb, 0)
    plan.generate_twiddles(1, g)
    for _ in range(3):
        plan.forward(x, y)
    best = min((plan.forward_profile(x, y) for _ in range(5)), key=sum)
    print(json.dumps({"logn": logn, "word_bytes": wb, "batch": batch, "dbg": flags, "pass_ms": [round(m, 3) for m in best]}), flush=True)
