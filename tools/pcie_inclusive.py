#!/usr/bin/env python3
"""PCIe-inclusive rate of the headline shape: the batch starts and ends in (pinned) HOST memory, as in the reference's host
procedure (bo.sync TO_DEVICE, run, bo.sync FROM_DEVICE: src/test.cpp:148-151, 168) -- chunks of polynomials pipelined over copy
and compute streams (H2D of chunk k+1, transform of chunk k, D2H of chunk k-1 overlap).  Never the bench's `value` (that one has
inputs resident in HBM); DESIGN.md quotes this number next to it.
usage: pcie_inclusive.py [--logn 16] [--batch 4096] [--chunk 256] [--reps 3]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ntt_aie_amd import NTTPlan

ap = argparse.ArgumentParser()
ap.add_argument("--logn", type=int, default=16)
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--chunk", type=int, default=256)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
p, g, n = 0xFFFFFFFF00000001, 7, 1 << a.logn
torch.cuda.set_device(0)
plan = NTTPlan(a.logn, p, 8, 0)
plan.generate_twiddles(0, g)
h_in = torch.randint(0, 1 << 62, (a.batch, n), dtype=torch.int64).pin_memory()
h_out = torch.empty_like(h_in).pin_memory()
nchunk = (a.batch + a.chunk - 1) // a.chunk
NBUF = 3
d_in = [torch.empty((a.chunk, n), dtype=torch.int64, device="cuda:0") for _ in range(NBUF)]
d_out = [torch.empty((a.chunk, n), dtype=torch.int64, device="cuda:0") for _ in range(NBUF)]
s_h2d, s_run, s_d2h = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
ev_in = [torch.cuda.Event() for _ in range(NBUF)]
ev_run = [torch.cuda.Event() for _ in range(NBUF)]
ev_out = [torch.cuda.Event() for _ in range(NBUF)]


def once():
    for k in range(nchunk):
        b = k % NBUF
        lo, hi = k * a.chunk, min(a.batch, (k + 1) * a.chunk)
        rows = hi - lo
        with torch.cuda.stream(s_h2d):
            s_h2d.wait_event(ev_run[b])  # the transform that last read this input buffer is done
            d_in[b][:rows].copy_(h_in[lo:hi], non_blocking=True)
            ev_in[b].record(s_h2d)
        with torch.cuda.stream(s_run):
            s_run.wait_event(ev_in[b])
            s_run.wait_event(ev_out[b])  # the copy-out that last read this output buffer is done
            plan.forward(d_in[b][:rows], d_out[b][:rows], stream=s_run)
            ev_run[b].record(s_run)
        with torch.cuda.stream(s_d2h):
            s_d2h.wait_event(ev_run[b])
            h_out[lo:hi].copy_(d_out[b][:rows], non_blocking=True)
            ev_out[b].record(s_d2h)
    torch.cuda.synchronize()


once()
# correctness of the pipeline: the same rows transformed directly
ref = plan.forward(h_in[:4].cuda())
assert torch.equal(ref.cpu(), h_out[:4]) and torch.equal(plan.forward(h_in[-4:].cuda()).cpu(), h_out[-4:])
best = None
for _ in range(a.reps):
    t0 = time.perf_counter()
    once()
    dt = time.perf_counter() - t0
    best = dt if best is None or dt < best else best
gib = a.batch * n * 8 / 2**30
print("PCIe-inclusive forward, N=2^%d batch %d (%.1f GiB in + %.1f GiB out through pinned host memory, chunks of %d, 3 streams): "
      "%.1f ms per batch = %.0f NTT/s, %.1f GB/s each way" % (a.logn, a.batch, gib, gib, a.chunk, best * 1e3, a.batch / best, gib * 2**30 / best / 1e9))
