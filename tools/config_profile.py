#!/usr/bin/env python3
"""One BASELINE configuration (tools/configs.py), looped: the program rocprofv3 wraps for kernel stats and counters of the
configurations OTHER than the headline (bench.py is the headline's).  usage: config_profile.py <cfg2|cfg2_sat|cfg3|cfg4> [reps]
Prints one line with the time per operation (hipEvents around the loop on the launch stream)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch

from configs import CONFIGS
from ntt_aie_amd import NTTPlan


def rand(batch, n, wb, p, seed):
    g = torch.Generator(device="cuda:0").manual_seed(seed)
    if wb == 8:
        hi = torch.randint(0, min(0xFFFFFFFF, max(1, p >> 32)), (batch, n), dtype=torch.int64, device="cuda:0", generator=g)  # canonical: < p
        lo = torch.randint(0, 1 << 32, (batch, n), dtype=torch.int64, device="cuda:0", generator=g)
        return (hi << 32) | lo
    return torch.randint(0, p, (batch, n), dtype=torch.int64, device="cuda:0", generator=g).to(torch.int32)


def main():
    c = CONFIGS[sys.argv[1]]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    n = 1 << c["logn"]
    plan = NTTPlan(c["logn"], c["p"], c["wb"], 0)
    plan.generate_twiddles(c["kind"], c["g"])
    stream = torch.cuda.current_stream()
    if c["op"] == "polymul":
        a, b = rand(c["batch"], n, c["wb"], c["p"], 2), rand(c["batch"], n, c["wb"], c["p"], 3)
        out = torch.empty_like(a)
        fn = lambda: plan.polymul_negacyclic(a, b, out, stream=stream)  # a, b are scratch: their contents stay canonical residues
    else:
        x = rand(c["batch"], n, c["wb"], c["p"], 1)
        y = torch.empty_like(x)
        fn = lambda: plan.forward(x, y, stream=stream)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    e1.synchronize()
    print("%s: %.4f ms per operation over %d (%s)" % (sys.argv[1], e0.elapsed_time(e1) / reps, reps, c["name"]))


if __name__ == "__main__":
    main()
