"""Diagnostics for the experimental fused launch (NTT_FUSED=1): correctness on sampled rows, time per
step and the control words of the last launch (slots per XCC, status, completed column tiles, verdict).
NTT_DEBUG_FLAGS=16/32/64 isolate the roles (see tools/fused_gl16.hip).  Usage: fused_diag.py BATCH"""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
# experiment knobs live only in libntt_hip_exp.so (make -C ntt_aie_amd/csrc exp): the product library reads no env
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _explib
_explib.select()  # the experiment build unless NTT_HIP_LIB names another one
os.environ["NTT_FUSED"] = "1"
import torch
from ntt_aie_amd import NTTPlan, to_device, to_host
p = 0xFFFFFFFF00000001; logn = 16; n = 1 << logn
batch = int(sys.argv[1])
plan = NTTPlan(logn, p, 8, 0); T = plan.make_roots(7); plan.set_twiddles(T)
a = np.random.default_rng(0).integers(0, 2**63, size=(batch, n), dtype=np.uint64) % np.uint64(p)
d = to_device(a, "cuda:0"); out = torch.zeros_like(d)
print("launch", flush=True)
plan.forward(d, out); torch.cuda.synchronize()
print("synced", flush=True)
rows = list(range(0, batch, max(1, batch // 16)))
# the checker is the library's own two-launch path (an unfused plan); tools/ never touch oracle/
os.environ["NTT_FUSED"] = "0"; ref_plan = NTTPlan(logn, p, 8, 0); ref_plan.set_twiddles(T); os.environ["NTT_FUSED"] = "1"
got = to_host(out[rows]); want = to_host(ref_plan.forward(d)[rows]) if not os.environ.get("NTT_DEBUG_FLAGS") else got
print("mismatching rows", int((got != want).any(axis=1).sum()), "of", len(rows))
for _ in range(3): plan.forward(d, out)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): plan.forward(d, out)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print("ms/step %.3f  NTT/s %.0f" % (dt * 1e3, batch / dt))
got = to_host(out[rows]); print("after timing: mismatching rows", int((got != want).any(axis=1).sum()))

from ntt_aie_amd import _lib
L = _lib.lib()
print("fused ctl: slots", [L.ntt_plan_info(plan._h, 16 + i) for i in range(8)], "status", L.ntt_plan_info(plan._h, 24),
      "b_done", L.ntt_plan_info(plan._h, 25), "want", batch * 16, "ok", L.ntt_plan_info(plan._h, 26))
