#!/usr/bin/env python3
"""Same-process, interleaved A/B of library builds on ONE launch shape of any word size: microseconds per launch of
ntt_forward (or ntt_inverse), K launches back to back between two events on the launch stream, variants alternated round
by round; every variant's output is compared word for word with the first one's.

usage: ab_latency.py [--logn 12] [--p 3221225473] [--g 5] [--word-bytes 4] [--batch 1024] [--rounds 9] [--k 50] [--inverse]
                     [--no-check] NAME=path[+ENV=VAL...] ...
  e.g. ab_latency.py base=ab/libntt_base.so new=ntt_aie_amd/libntt_hip.so"""
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
ap.add_argument("--logn", type=int, default=12)
ap.add_argument("--p", type=int, default=3221225473)
ap.add_argument("--g", type=int, default=5)
ap.add_argument("--word-bytes", type=int, default=4)
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--rounds", type=int, default=9)
ap.add_argument("--k", type=int, default=50)
ap.add_argument("--inverse", action="store_true")
ap.add_argument("--no-check", action="store_true", help="timing experiments whose outputs are meaningless (NTT_DEBUG_FLAGS)")
ap.add_argument("variants", nargs="+")
args = ap.parse_args()

torch.cuda.set_device(0)
n = 1 << args.logn
gen = torch.Generator(device="cuda:0").manual_seed(7)
if args.word_bytes == 4:
    x = torch.randint(0, args.p, (args.batch, n), dtype=torch.int64, device="cuda:0", generator=gen).to(torch.int32)
else:
    x = torch.randint(0, 1 << 62, (args.batch, n), dtype=torch.int64, device="cuda:0", generator=gen)
stream = torch.cuda.current_stream()
plans = []
for v in args.variants:
    name, rest = v.split("=", 1)
    parts = rest.split("+")  # path[+ENV=VAL...]: set while this variant's plan is created (experiment builds read knobs there)
    path = parts[0] if os.path.isabs(parts[0]) else os.path.join(ROOT, parts[0])
    env = dict(q.split("=", 1) for q in parts[1:])
    try:
        L = _lib.open_library(path)
    except AttributeError:  # a build from before round 3 (no ntt_plan_select / _set_policy / _clone)
        L = _lib.open_library(path, since_v3=False)
    h = C.c_void_p()
    old_env = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    rc = L.ntt_plan_create(C.byref(h), args.logn, args.p, args.word_bytes, 0)
    for k, o in old_env.items():
        if o is None:
            del os.environ[k]
        else:
            os.environ[k] = o
    assert rc == 0, name
    assert L.ntt_plan_generate_twiddles(h, 0, args.g) == 0, name
    plans.append((name, L, h, None))


def launch(L, h, y):
    if args.inverse:
        rc = L.ntt_inverse(h, x.data_ptr(), y.data_ptr(), args.batch, 0, 1, stream.cuda_stream)
    else:
        rc = L.ntt_forward(h, x.data_ptr(), y.data_ptr(), args.batch, 0, stream.cuda_stream)
    assert rc == 0, rc


def timed(L, h, y):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(args.k):
        launch(L, h, y)
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / args.k


# ONE output buffer for every variant (where a buffer lies in HBM is worth up to +-5 % on a memory-bound launch: with a buffer
# per variant, identical kernels measured that far apart); each variant's output is checked against the first one's copy
y = torch.empty_like(x)
plans = [(name, L, h, y) for name, L, h, _ in plans]
same, ref = True, None
for name, L, h, _ in plans:
    for _ in range(20):
        launch(L, h, y)
    torch.cuda.synchronize()
    if ref is None:
        ref = y.clone()
    else:
        same = same and torch.equal(y, ref)
del ref
assert same or args.no_check, "a variant's output differs from %s's" % plans[0][0]
samples = {name: [] for name, _, _, _ in plans}
for r in range(args.rounds):
    for name, L, h, y in plans:
        samples[name].append(timed(L, h, y))
print("logn=%d p=%d word_bytes=%d batch=%d %s: us per launch (%d launches back to back, %d interleaved rounds); outputs %s" % (
    args.logn, args.p, args.word_bytes, args.batch, "inverse" if args.inverse else "forward", args.k, args.rounds, "identical" if same else "DIFFER (--no-check)"))
for name, _, _, _ in plans:
    s = samples[name]
    print("  %-12s median %8.3f  min %8.3f" % (name, statistics.median(s), min(s)), flush=True)
