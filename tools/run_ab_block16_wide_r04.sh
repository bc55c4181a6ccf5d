#!/bin/bash
# AIE_BLOCK16 straight from the radix-8 kernels (pass.h: elem_off / lane_eff): the commit before (ab/libntt_prev.so: the layout kept
# the radix-16 kernel) against this one.  (1) no regression where the 512-thread kernels run as the FIRST pass (N = 2^18, 2^20 forward);
# (2) the reference's own launch shape through the layout: N = 2^11, p = 3329, batch 1 -- ab_latency times the natural layout, so the
# layout legs are timed by tools/archive/one_size.py-style loops below.
set -e
cd "$GRAFT_REPO_ROOT"
G=18446744069414584321
for N in 18 20; do
  python3 tools/ab_latency.py --logn $N --p $G --g 7 --word-bytes 8 --batch 256 --rounds 5 --k 10 prev=ab/libntt_prev.so new=ntt_aie_amd/libntt_hip.so 2>&1 | grep -v amdgpu.ids
done
python3 - <<'PY'
import ctypes as C, os, statistics, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from ntt_aie_amd import _lib
torch.cuda.set_device(0)
s = torch.cuda.current_stream()
for logn, p, g, wb in ((11, 3329, 3, 4), (12, 3221225473, 5, 4), (12, 0xFFFFFFFF00000001, 7, 8)):
    n = 1 << logn
    x = (torch.arange(n, dtype=torch.int64, device="cuda:0") % p).to(torch.int32 if wb == 4 else torch.int64)[None, :].contiguous()
    res = {}
    outs = {}
    for name, path in (("prev", "ab/libntt_prev.so"), ("new", "ntt_aie_amd/libntt_hip.so")):
        L = _lib.open_library(os.path.join(os.getcwd(), path))
        h = C.c_void_p()
        assert L.ntt_plan_create(C.byref(h), logn, p, wb, 0) == 0
        assert L.ntt_plan_generate_twiddles(h, 0, g) == 0
        y = torch.empty_like(x)
        t = []
        for r in range(7):
            for _ in range(5):
                L.ntt_forward(h, x.data_ptr(), y.data_ptr(), 1, 1, s.cuda_stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(50):
                L.ntt_forward(h, x.data_ptr(), y.data_ptr(), 1, 1, s.cuda_stream)   # layout 1 = AIE_BLOCK16
            e1.record(s); e1.synchronize()
            t.append(e0.elapsed_time(e1) / 50 * 1e3)
        res[name] = statistics.median(t)
        outs[name] = y.clone()
    print("N=2^%d p=%d %d-byte batch 1 AIE_BLOCK16 forward: prev %.3f us  new %.3f us (%+.1f %%)  outputs %s" % (
        logn, p, wb, res["prev"], res["new"], 100 * (res["new"] / res["prev"] - 1), "identical" if torch.equal(outs["prev"], outs["new"]) else "DIFFER"))
PY
