#!/usr/bin/env python3
"""Same-process A/B of ntt_pointwise_mul across library builds (one set of buffers for every variant, outputs compared):
usage: ab_pointwise.py NAME=path ...   (1 GiB per operand, both word sizes, with and without the scale factor)"""
import ctypes as C, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench_configs as B
from ntt_aie_amd import _lib

stream = torch.cuda.current_stream()
for wb, p in ((8, B.GOLD), (4, 998244353), (4, 3221225473)):
    n, logn = 4096, 12
    batch = (1 << 30) // (n * wb)
    a, b = B.rand(batch, n, wb, p, 2), B.rand(batch, n, wb, p, 3)
    c = torch.empty_like(a)
    libs = []
    for v in sys.argv[1:]:
        name, path = v.split("=", 1)
        L = _lib.open_library(path if os.path.isabs(path) else os.path.join(ROOT, path))
        h = C.c_void_p()
        assert L.ntt_plan_create(C.byref(h), logn, p, wb, 0) == 0
        libs.append((name, L, h))
    for scale in (1, 12345):
        ref, same, res = None, True, {}
        for name, L, h in libs:
            assert L.ntt_pointwise_mul(h, a.data_ptr(), b.data_ptr(), c.data_ptr(), batch, scale, stream.cuda_stream) == 0
            torch.cuda.synchronize()
            if ref is None: ref = c.clone()
            else: same = same and torch.equal(c, ref)
        for r in range(7):
            for name, L, h in libs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(10):
                    L.ntt_pointwise_mul(h, a.data_ptr(), b.data_ptr(), c.data_ptr(), batch, scale, stream.cuda_stream)
                e1.record(stream); e1.synchronize()
                res.setdefault(name, []).append(e0.elapsed_time(e1) * 100)
        print("word_bytes=%d p=%d scale=%d: us per call, 3 GiB of traffic (copy rate: 565 us); outputs %s" % (wb, p, scale, "identical" if same else "DIFFER"))
        for name, _, _ in libs:
            print("  %-8s median %8.1f  min %8.1f" % (name, statistics.median(res[name]), min(res[name])), flush=True)
    del a, b, c
