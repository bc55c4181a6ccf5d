#!/usr/bin/env python3
"""Per-pass kernel times (ntt_forward_profile: hipEvents around every pass) of one decomposition at several batches.
usage: pass_times.py --logn 22 --word-bytes 4 --p 998244353 --g 3 --split 13,9 --batches 8,16,32,64,128  (experiment build: NTT_PLAN_SPLIT)"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ntt_aie_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--logn", type=int, default=22)
ap.add_argument("--word-bytes", type=int, default=4)
ap.add_argument("--p", type=int, default=998244353)
ap.add_argument("--g", type=int, default=3)
ap.add_argument("--split", default="")
ap.add_argument("--batches", default="8,16,32,64,128")
ap.add_argument("--reps", type=int, default=7)
a = ap.parse_args()
L = _lib.open_library(os.path.join(ROOT, "ntt_aie_amd", "libntt_hip_exp.so"))
if a.split:
    os.environ["NTT_PLAN_SPLIT"] = a.split
h = C.c_void_p()
assert L.ntt_plan_create(C.byref(h), a.logn, a.p, a.word_bytes, 0) == 0
assert L.ntt_plan_generate_twiddles(h, 0, a.g) == 0
torch.cuda.set_device(0)
st = torch.cuda.current_stream()
n = 1 << a.logn
for b in map(int, a.batches.split(",")):
    x = torch.randint(0, min(a.p, 1 << 62), (b, n), dtype=torch.int64, device="cuda:0")
    if a.word_bytes == 4:
        x = x.to(torch.int32)
    y = torch.empty_like(x)
    ms, k = (C.c_float * 8)(), C.c_int(0)
    best = None
    for _ in range(a.reps):
        assert L.ntt_forward_profile(h, x.data_ptr(), y.data_ptr(), b, 0, st.cuda_stream, ms, 8, C.byref(k)) == 0
        cur = [ms[i] for i in range(k.value)]
        best = cur if best is None or sum(cur) < sum(best) else best
    print("split %-8s batch %4d  per pass us: %s  total %.1f  per poly %.2f" % (a.split or "plan", b, " ".join("%8.1f" % (v * 1e3) for v in best), sum(best) * 1e3, sum(best) * 1e3 / b), flush=True)
    del x, y
