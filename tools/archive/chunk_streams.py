#!/usr/bin/env python3
"""Experiment: forward transform of the headline batch as CHUNKS on several streams inside one hipGraph, so that the
second pass of a chunk reads its intermediate from the 256 MiB Infinity Cache instead of HBM while another stream
runs the first pass of the next chunk (the board is power-capped: fewer HBM bytes = higher clock).
usage: chunk_streams.py [chunk ...]   (default 64 128 256 512 1024)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import GOLDILOCKS, synth_batch  # noqa: E402
from ntt_aie_amd import NTTPlan  # noqa: E402

torch.cuda.set_device(0)
logn, batch = 16, 4096
n = 1 << logn
plan = NTTPlan(logn, GOLDILOCKS, 8, 0)
plan.generate_twiddles(0, 7)
x = synth_batch(torch, batch, n, torch.device("cuda", 0))
y = torch.empty_like(x)
ref = plan.forward(x)
torch.cuda.synchronize()


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print("baseline (two launches)      %.4f ms" % timed(lambda: plan.forward(x, y)), flush=True)
g0 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g0):
    plan.forward(x, y)
print("baseline in a graph          %.4f ms" % timed(g0.replay), flush=True)
chunks = [int(a) for a in sys.argv[1:]] or [64, 128, 256, 512, 1024]
for nstreams in (1, 2, 3):
    for chunk in chunks:
        streams = [torch.cuda.Stream() for _ in range(nstreams)]
        y.zero_()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            main = torch.cuda.current_stream()
            for s in streams:
                s.wait_stream(main)
            for i, c0 in enumerate(range(0, batch, chunk)):
                s = streams[i % nstreams]
                with torch.cuda.stream(s):
                    plan.forward(x[c0:c0 + chunk], y[c0:c0 + chunk], stream=s)
            for s in streams:
                main.wait_stream(s)
        ms = timed(g.replay)
        ok = torch.equal(y, ref)
        print("streams %d chunk %4d          %.4f ms  %s" % (nstreams, chunk, ms, "ok" if ok else "MISMATCH"), flush=True)
