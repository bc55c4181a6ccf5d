#!/usr/bin/env python3
"""the MI355X series in the reference's own profile/ data formats, so that its plots can lay this
engine beside the AIE and the A100 (profile/plot_efficiency.py:25-27,44-46, profile/plot_exectime.py:27-29).

  profiles/kerneltime/mi355x.csv          "N , kernel microseconds" rows, N = 2^8 .. 2^17, batch 1 (the axis of
                                          profile/kerneltime/gpu.csv); kernel time = sum of the transform's pass kernels
                                          between hipEvents (ntt_forward_profile), median of 30 launches; p = 3329, g = 3,
                                          4-byte words, a[i] = i mod p (the reference's parameter set, src/test.cpp:66,76-77)
  profiles/kerneltime/mi355x_batch.csv    the same sizes at a saturating batch, microseconds PER TRANSFORM
  profiles/exectime/ntt_mi355x_logn{8..13}.csv   the reference's 10-launch test procedure (host.reference_procedure: launch +
                                          wait, wall clock, one integer microsecond per line), verified against the oracle

Lives under tests/ because it uses the oracle as the checker of the reference procedure (only tests/, smoke() and bench.py's
cpu_baseline may touch oracle/).  Run on the GPU box: python3 tests/profile_series.py [outdir]   (default gpurun_out/profiles_series; copy into profiles/)."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import io  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle_py as O  # noqa: E402  (checker only: the expected words of the reference procedure)
from ntt_aie_amd import NTTPlan, host, to_device  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "profiles_series")
os.makedirs(os.path.join(out, "kerneltime"), exist_ok=True)
os.makedirs(os.path.join(out, "exectime"), exist_ok=True)
torch.cuda.set_device(0)
P, G = 3329, 3

rows1, rowsb = [], []
for logn in range(8, 18):
    n = 1 << logn
    plan = NTTPlan(logn, P, 4, 0)
    plan.set_twiddles(plan.make_roots(G))
    for batch, rows in ((1, rows1), (max(1, (1 << 28) >> logn), rowsb)):  # 1 GiB of coefficients at the saturating batch
        a = (np.arange(n, dtype=np.uint64) % P).astype(np.uint32)
        x = to_device(np.broadcast_to(a, (batch, n)).copy(), "cuda:0")
        y = torch.empty_like(x)
        for _ in range(5):
            plan.forward(x, y)
        us = statistics.median(sum(plan.forward_profile(x, y)) * 1e3 for _ in range(30))
        rows.append(host.kerneltime_row(n, us / batch))
        del x, y
with open(os.path.join(out, "kerneltime", "mi355x.csv"), "w") as f:
    f.write("\n".join(rows1) + "\n")
with open(os.path.join(out, "kerneltime", "mi355x_batch.csv"), "w") as f:
    f.write("\n".join(rowsb) + "\n")
print("kerneltime batch 1:", rows1)
print("kerneltime per transform at the saturating batch:", rowsb)

for logn in range(8, 14):
    n = 1 << logn
    want = O.ntt(np.arange(n, dtype=np.uint32) % P, O.make_roots(n, P, G, 4), P)
    buf = io.StringIO()
    rc, times = host.reference_procedure(logn=logn, p=P, g=G, expected_natural=want, out=buf)
    assert rc == 0, buf.getvalue()
    host.write_exectime_csv(os.path.join(out, "exectime", "ntt_mi355x_logn%d.csv" % logn), times)
    print("exectime logn %d: trimmed mean %.1f us (PASS)" % (logn, host.trimmed_mean(times)))
