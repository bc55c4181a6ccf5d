#!/usr/bin/env python3
"""Same-process, interleaved A/B of library builds (methodology: N variants x M rounds in ONE process, one device):
per-pass kernel times of the forward transform from ntt_forward_profile (hipEvents around every pass).

usage: ab_pass.py [--logn 16] [--batch 4096] [--rounds 7] [--reps 5] [--dbg FLAGS] [--inverse] NAME=path[+ENV=VAL...] ...
  e.g. ab_pass.py base=ab/libntt_base.so new=ntt_aie_amd/libntt_hip.so
       ab_pass.py --dbg 3 exp=ntt_aie_amd/libntt_hip_exp.so     (VALU floor; experiment builds only)
+ENV=VAL pairs (e.g. +NTT_PLAN_SPLIT=9,7) are set while that variant's plan is created (experiment builds read their knobs there).
Prints, per variant: median and min of the per-pass times and of their sum over all rounds."""
import argparse
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import GOLDILOCKS, synth_batch  # noqa: E402
from ntt_aie_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--logn", type=int, default=16)
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--dbg", default=None)
ap.add_argument("variants", nargs="+")
args = ap.parse_args()

torch.cuda.set_device(0)
n = 1 << args.logn
x = synth_batch(torch, args.batch, n, torch.device("cuda", 0))
y = torch.empty_like(x)
stream = torch.cuda.current_stream()
plans = []
for v in args.variants:
    name, rest = v.split("=", 1)
    parts = rest.split("+")
    path = parts[0] if os.path.isabs(parts[0]) else os.path.join(ROOT, parts[0])
    env = dict(p.split("=", 1) for p in parts[1:])
    if args.dbg is not None:
        env["NTT_DEBUG_FLAGS"] = args.dbg
    L = _lib.open_library(path)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    h = C.c_void_p()
    rc = L.ntt_plan_create(C.byref(h), args.logn, GOLDILOCKS, 8, 0)
    for k, o in old.items():
        if o is None:
            del os.environ[k]
        else:
            os.environ[k] = o
    assert rc == 0, (name, rc)
    assert L.ntt_plan_generate_twiddles(h, 0, 7) == 0
    plans.append((name, L, h))

ms, k = (C.c_float * 8)(), C.c_int(0)


def run(L, h):
    rc = L.ntt_forward_profile(h, x.data_ptr(), y.data_ptr(), args.batch, 0, stream.cuda_stream, ms, 8, C.byref(k))
    assert rc == 0, rc
    return [float(ms[i]) for i in range(k.value)]


for name, L, h in plans:  # warm-up
    for _ in range(4):
        run(L, h)
samples = {name: [] for name, _, _ in plans}
for r in range(args.rounds):
    for name, L, h in plans:
        for _ in range(args.reps):
            samples[name].append(run(L, h))
for name, _, _ in plans:
    s = samples[name]
    np_ = len(s[0])
    med = [statistics.median(v[i] for v in s) for i in range(np_)]
    mn = [min(v[i] for v in s) for i in range(np_)]
    tot = [sum(v) for v in s]
    print("%-12s passes median %s  min %s   sum median %.4f  min %.4f ms" % (
        name, ["%.4f" % m for m in med], ["%.4f" % m for m in mn], statistics.median(tot), min(tot)), flush=True)
