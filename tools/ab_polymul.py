#!/usr/bin/env python3
"""Same-process, interleaved A/B of ntt_polymul_negacyclic across library builds (kind-2 table made on the device); every variant's
product is compared word for word with the first one's.
usage: ab_polymul.py [--logn 16] [--p P --g G] [--word-bytes 8] [--batch 4096] [--k 5] [--rounds 5] NAME=path ..."""
import argparse, ctypes as C, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ntt_aie_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--logn", type=int, default=16)
ap.add_argument("--p", type=lambda v: int(v, 0), default=0xFFFFFFFF00000001)
ap.add_argument("--g", type=int, default=7)
ap.add_argument("--word-bytes", type=int, default=8)
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--k", type=int, default=5)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("variants", nargs="+")
a = ap.parse_args()
torch.cuda.set_device(0)
n = 1 << a.logn
gen = torch.Generator(device="cuda:0").manual_seed(3)
hi = min(a.p, 1 << 62)
mk = lambda: torch.randint(0, hi, (a.batch, n), dtype=torch.int64, device="cuda:0", generator=gen).to(torch.int32 if a.word_bytes == 4 else torch.int64)
x, y = mk(), mk()
wa, wb, out = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
st = torch.cuda.current_stream()
plans = []
for v in a.variants:
    name, path = v.split("=", 1)
    path = path if os.path.isabs(path) else os.path.join(ROOT, path)
    try:
        L = _lib.open_library(path)
    except AttributeError:
        L = _lib.open_library(path, since_v3=False)
    h = C.c_void_p()
    assert L.ntt_plan_create(C.byref(h), a.logn, a.p, a.word_bytes, 0) == 0, name
    assert L.ntt_plan_generate_twiddles(h, 2, a.g) == 0, name
    plans.append((name, L, h))


def run(L, h):
    wa.copy_(x)
    wb.copy_(y)  # the product overwrites its operands
    assert L.ntt_polymul_negacyclic(h, wa.data_ptr(), wb.data_ptr(), out.data_ptr(), a.batch, st.cuda_stream) == 0


ref, same = None, True
for name, L, h in plans:
    for _ in range(3):
        run(L, h)
    torch.cuda.synchronize()
    if ref is None:
        ref = out.clone()
    else:
        same = same and torch.equal(out, ref)
samples = {name: [] for name, _, _ in plans}
for _ in range(a.rounds):
    for name, L, h in plans:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(a.k):
            run(L, h)
        e1.record(st)
        e1.synchronize()
        samples[name].append(e0.elapsed_time(e1) * 1e3 / a.k)
print("polymul logn=%d p=%#x word_bytes=%d batch=%d: us per product batch incl. two operand copies (%d back to back, %d interleaved rounds); outputs %s" % (
    a.logn, a.p, a.word_bytes, a.batch, a.k, a.rounds, "identical" if same else "DIFFER"))
for name, _, _ in plans:
    print("  %-12s median %10.1f  min %10.1f" % (name, statistics.median(samples[name]), min(samples[name])), flush=True)
