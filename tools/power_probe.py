#!/usr/bin/env python3
"""Board power and clocks (rocm-smi, sampled from a side thread) while one workload loops for a few seconds each:
the real forward transform, the same kernels with L2-resident loads and no stores (experiment build), and a plain copy.
Shows whether the transform runs against the board's power cap (the clock it holds then is what sets its speed)."""
import ctypes as C
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import GOLDILOCKS, synth_batch  # noqa: E402
from ntt_aie_amd import _lib  # noqa: E402

SECONDS = float(os.environ.get("PROBE_SECONDS", "4"))
torch.cuda.set_device(0)
x = synth_batch(torch, 4096, 1 << 16, torch.device("cuda", 0))
y = torch.empty_like(x)
exp = _lib.open_library(os.path.join(ROOT, "ntt_aie_amd", "libntt_hip_exp.so"))


def plan(dbg, only_pass=None):
    if dbg:
        os.environ["NTT_DEBUG_FLAGS"] = str(dbg)
    if only_pass is not None:
        os.environ["NTT_ONLY_PASS"] = str(only_pass)
    h = C.c_void_p()
    assert exp.ntt_plan_create(C.byref(h), 16, GOLDILOCKS, 8, 0) == 0
    os.environ.pop("NTT_DEBUG_FLAGS", None)
    os.environ.pop("NTT_ONLY_PASS", None)
    assert exp.ntt_plan_generate_twiddles(h, 0, 7) == 0
    return h


def smi():
    out = subprocess.run(["rocm-smi", "-d", "0", "--showpower", "--showclocks", "--showmaxpower", "--showtemp"], capture_output=True, text=True).stdout
    pw = re.search(r"(?:Average|Current Socket) Graphics Package Power \(W\): ([\d.]+)", out)
    cap = re.search(r"Max Graphics Package Power \(W\): ([\d.]+)", out)
    sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
    mclk = re.search(r"mclk clock level: \d+: \((\d+)Mhz\)", out)
    return (float(pw.group(1)) if pw else None, float(cap.group(1)) if cap else None,
            int(sclk.group(1)) if sclk else None, int(mclk.group(1)) if mclk else None, out)


def probe(name, fn):
    stop = [False]
    samples = []

    def sampler():
        while not stop[0]:
            samples.append(smi()[:4])
            time.sleep(0.2)

    t = threading.Thread(target=sampler)
    fn(); torch.cuda.synchronize()
    t.start()
    t0 = time.perf_counter(); it = 0
    while time.perf_counter() - t0 < SECONDS:
        for _ in range(50):
            fn()
        torch.cuda.synchronize(); it += 50
    dt = time.perf_counter() - t0
    stop[0] = True; t.join()
    pw = [s[0] for s in samples if s[0] is not None]
    sc = [s[2] for s in samples if s[2] is not None]
    print("%-28s %.4f ms/iter  power W: max %s mean %s cap %s  sclk MHz: %s  mclk %s" % (
        name, dt / it * 1e3, max(pw) if pw else None, round(sum(pw) / len(pw), 1) if pw else None,
        samples[-1][1] if samples else None, sorted(set(sc)), samples[-1][3] if samples else None), flush=True)


print(smi()[4][:1500])
s = torch.cuda.current_stream().cuda_stream
h0, h3 = plan(0), plan(3)
probe("idle-ish (sync only)", lambda: None)
probe("forward (real)", lambda: exp.ntt_forward(h0, x.data_ptr(), y.data_ptr(), 4096, 0, s))
probe("forward (L2 loads, no stores)", lambda: exp.ntt_forward(h3, x.data_ptr(), y.data_ptr(), 4096, 0, s))
probe("copy (xor kernel)", lambda: torch.bitwise_xor(x, 1, out=y))
probe("forward (real) again", lambda: exp.ntt_forward(h0, x.data_ptr(), y.data_ptr(), 4096, 0, s))
# each pass kernel alone (NTT_ONLY_PASS, experiment build): does one of the two cost more power per byte than the other?
hp0, hp1 = plan(0, 0), plan(0, 1)
probe("CONTIG pass alone (stages 0-7)", lambda: exp.ntt_forward(hp0, x.data_ptr(), y.data_ptr(), 4096, 0, s))
probe("column pass alone (stages 8-15)", lambda: exp.ntt_forward(hp1, x.data_ptr(), y.data_ptr(), 4096, 0, s))
hf0, hf1 = plan(3, 0), plan(3, 1)
probe("CONTIG pass alone, L2 loads, no stores", lambda: exp.ntt_forward(hf0, x.data_ptr(), y.data_ptr(), 4096, 0, s))
probe("column pass alone, L2 loads, no stores", lambda: exp.ntt_forward(hf1, x.data_ptr(), y.data_ptr(), 4096, 0, s))
