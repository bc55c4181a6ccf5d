#!/usr/bin/env python3
"""Same-process regression sweep of two library builds over EVERY size: logn 1..LOGN_MAX x {batch 1, the batch that makes
4 GiB (--bytes)} x {forward, inverse} for one field; interleaved rounds, ONE output buffer, outputs compared word for word.
Prints one line per shape with both medians and the delta, then the shapes where NEW is slower than OLD by more than --tol.

usage: regress_sweep.py [--word-bytes 8] [--p P --g G] [--max-logn 24] [--bytes 4294967296] [--tol 0.03] OLD.so NEW.so
  e.g. regress_sweep.py ab/libntt_r02.so ntt_aie_amd/libntt_hip.so          (tools/ab_build_rev.sh r02 <rev> makes the old one)"""
import argparse
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from ntt_aie_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--word-bytes", type=int, default=8)
ap.add_argument("--p", type=int, default=0xFFFFFFFF00000001)
ap.add_argument("--g", type=int, default=7)
ap.add_argument("--min-logn", type=int, default=1)
ap.add_argument("--max-logn", type=int, default=24)
ap.add_argument("--bytes", type=int, default=1 << 32)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--tol", type=float, default=0.03)
ap.add_argument("--batches", default="1,full", help="comma list: integers and/or 'full' (= --bytes worth of polynomials)")
ap.add_argument("libs", nargs=2)
args = ap.parse_args()

torch.cuda.set_device(0)
stream = torch.cuda.current_stream()
libs = [_lib.open_library(p if os.path.isabs(p) else os.path.join(ROOT, p), since_v3=(i == 1)) for i, p in enumerate(args.libs)]
wb = args.word_bytes
slower = []
print("# %s (OLD) vs %s (NEW); word_bytes=%d p=%d; us per launch, median of %d interleaved rounds" % (args.libs[0], args.libs[1], wb, args.p, args.rounds))
for logn in range(args.min_logn, args.max_logn + 1):
    n = 1 << logn
    full = max(1, args.bytes // (n * wb))
    for bspec in args.batches.split(","):
        batch = full if bspec == "full" else int(bspec)
        if bspec != "full" and batch >= full:
            continue
        gen = torch.Generator(device="cuda:0").manual_seed(logn)
        if wb == 4:
            x = torch.randint(0, args.p, (batch, n), dtype=torch.int64, device="cuda:0", generator=gen).to(torch.int32)
        else:
            x = torch.randint(0, 1 << 62, (batch, n), dtype=torch.int64, device="cuda:0", generator=gen)
        y = torch.empty_like(x)
        plans = []
        for L in libs:
            h = C.c_void_p()
            assert L.ntt_plan_create(C.byref(h), logn, args.p, wb, 0) == 0
            assert L.ntt_plan_generate_twiddles(h, 0, args.g) == 0
            plans.append(h)
        k = 3 if batch * n * wb >= (1 << 30) else (10 if batch * n * wb >= (1 << 26) else 50)
        for inverse in (0, 1):
            def launch(i):
                L, h = libs[i], plans[i]
                rc = (L.ntt_inverse(h, x.data_ptr(), y.data_ptr(), batch, 0, 1, stream.cuda_stream) if inverse else
                      L.ntt_forward(h, x.data_ptr(), y.data_ptr(), batch, 0, stream.cuda_stream))
                assert rc == 0, rc

            outs = []
            for i in range(2):
                for _ in range(3):
                    launch(i)
                torch.cuda.synchronize()
                outs.append(y.clone() if i == 0 else None)
                if i == 1:
                    same = torch.equal(y, outs[0])
            outs = None
            t = [[], []]
            for _ in range(args.rounds):
                for i in range(2):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    for _ in range(k):
                        launch(i)
                    e1.record(stream)
                    e1.synchronize()
                    t[i].append(e0.elapsed_time(e1) * 1e3 / k)
            m0, m1 = statistics.median(t[0]), statistics.median(t[1])
            d = m1 / m0 - 1
            tag = "" if same else "  OUTPUTS DIFFER"
            print("logn %2d batch %8d %s  old %10.2f  new %10.2f  %+6.1f %%%s" % (logn, batch, "inv" if inverse else "fwd", m0, m1, 100 * d, tag), flush=True)
            if d > args.tol or not same:
                slower.append((logn, batch, "inv" if inverse else "fwd", m0, m1, d, same))
        for L, h in zip(libs, plans):
            L.ntt_plan_destroy(h)
        del x, y
print("# shapes where NEW is slower than OLD by more than %.0f %% (or differs): %d" % (100 * args.tol, len(slower)))
for s in slower:
    print("#   logn %d batch %d %s: %.2f -> %.2f us (%+.1f %%)%s" % (s[0], s[1], s[2], s[3], s[4], 100 * s[5], "" if s[6] else " OUTPUTS DIFFER"))
