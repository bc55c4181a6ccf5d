"""The reference's own test procedure, driven through the MI355X engine.

Mirrors src/test.cpp:137-247 step by step -- table by make_roots, input a[i] = i, ten timed
launches printed in microseconds, one more launch, word-by-word comparison in the device's
block order, PASS/FAIL and exit code -- and the data formats of the reference's profile/
directory (one launch time per line; "N , kernel_us" rows; the 5.5*N*log2(N) operation count
of profile/plot_efficiency.py), so its numbers can be laid beside the reference's.

The expected words are supplied by the caller (the tests compute them with the oracle): the
product itself has no CPU implementation of the transform.
"""
from __future__ import annotations

import math
import sys
import time

import numpy as np
import torch

from .plan import LAYOUT_AIE_BLOCK16, NTTPlan, to_device, to_host

# src/test.cpp:69-71 (data): device block ans_order[i] holds natural-order block i
BLOCK_NUM = 16
ANS_ORDER = (0, 2, 1, 3, 8, 10, 9, 11, 4, 6, 5, 7, 12, 14, 13, 15)


def block_order(a_ref: np.ndarray) -> np.ndarray:
    """src/test.cpp:212-219: answers[ans_order[i]*bs + j] = a_ref[i*bs + j]."""
    n = a_ref.shape[-1]
    bs = n // BLOCK_NUM
    answers = np.empty_like(a_ref)
    for i in range(BLOCK_NUM):
        answers[..., ANS_ORDER[i] * bs:(ANS_ORDER[i] + 1) * bs] = a_ref[..., i * bs:(i + 1) * bs]
    return answers


def reference_procedure(logn: int = 11, p: int = 3329, g: int = 3, expected_natural: np.ndarray | None = None,
                        launches: int = 10, device: int = 0, out=sys.stdout) -> tuple[int, list[float]]:
    """Run the reference's test (defaults = its compile-time constants, src/test.cpp:66, 76-77).

    Returns (exit_code, launch_times_us); exit_code 0 = PASS, 1 = FAIL like src/test.cpp:240-247.
    The reference always verifies before it prints PASS (src/test.cpp:212-247); without `expected_natural` nothing can
    be compared, so the procedure then prints NOT VERIFIED and returns 2 -- never PASS.
    """
    n = 1 << logn
    plan = NTTPlan(logn, p, 4, device)
    root = plan.make_roots(g)                                   # test.cpp:137-139
    buf_in = (np.arange(n, dtype=np.uint64) % p).astype(np.uint32)[None, :]  # test.cpp:141 (i < p there)
    plan.set_twiddles(root)                                     # bo_root.sync(TO_DEVICE), :150
    d_in = to_device(buf_in, "cuda:%d" % device)                # bo_inA.sync, :149
    d_out = torch.zeros_like(d_in)                              # bufOut[i] = 0, :143
    stream = torch.cuda.current_stream()
    print("Running Kernel.", file=out)
    times = []
    for _ in range(launches):                                   # test.cpp:157-175
        start = time.perf_counter()
        plan.forward(d_in, d_out, layout=LAYOUT_AIE_BLOCK16, stream=stream)
        stream.synchronize()                                    # run.wait()
        stop = time.perf_counter()
        buf_out = to_host(d_out)                                # bo_outC.sync(FROM_DEVICE), after the stop stamp
        us = (stop - start) * 1e6
        times.append(us)
        print(int(us), file=out)
    plan.forward(d_in, d_out, layout=LAYOUT_AIE_BLOCK16, stream=stream)  # test.cpp:180-190
    stream.synchronize()
    buf_out = to_host(d_out)[0]
    print("=================================", file=out)
    print("Verifying results", file=out)
    print("  logN: %d" % logn, file=out)
    print("  p: %d" % p, file=out)
    if expected_natural is None:
        print("  NOT VERIFIED (no expected words supplied).\n", file=out)
        return 2, times
    answers = block_order(np.asarray(expected_natural, dtype=np.uint32))  # test.cpp:212-219
    errors = int(np.count_nonzero(answers != buf_out))                    # test.cpp:224-235
    if not errors:
        print("  PASS!", file=out)
        return 0, times
    print("  mismatches: %d" % errors, file=out)
    print("  FAIL.\n", file=out)
    return 1, times


# ---- profile/ formats -----------------------------------------------------------------
def write_exectime_csv(path: str, times_us) -> None:
    """profile/exectime/ntt_*core_logn*.csv: one launch time (integer microseconds) per line."""
    with open(path, "w") as f:
        for t in times_us:
            f.write("%d\n" % int(t))


def trimmed_mean(times_us) -> float:
    """profile/plot_exectime.py:27-29: drop every sample equal to the max or the min, then mean."""
    a = np.asarray(times_us, dtype=float)
    kept = a[(a != a.max()) & (a != a.min())]
    return float(kept.mean()) if kept.size else float(a.mean())


def kerneltime_row(n: int, kernel_us: float) -> str:
    """profile/kerneltime/{aie,gpu}.csv row: "N , microseconds"."""
    return "%d , %.5f" % (n, kernel_us)


def efficiency(n: int, kernel_us: float, peak_gops: float) -> float:
    """profile/plot_efficiency.py:25-27, 44-46: 5.5*N*log2(N) operations over peak GOPS."""
    ops = 5.5 * n * math.log2(n)
    return ops / (kernel_us * 1e-6) / (peak_gops * 1e9)
