import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import torch, oracle_py as O
from ntt_aie_amd import NTTPlan, to_device, to_host
p = O.GOLDILOCKS
logn = int(sys.argv[1]); batch = int(sys.argv[2])
n = 1 << logn
plan = NTTPlan(logn, p, 8, 0); T = plan.make_roots(7); plan.set_twiddles(T)
a = np.random.default_rng(0).integers(0, 2**63, size=(batch, n), dtype=np.uint64) % np.uint64(p)
d = to_device(a, "cuda:0")
out = torch.empty_like(d)
print("in %x out %x bytes %x" % (d.data_ptr(), out.data_ptr(), d.numel()*8), flush=True)
f = plan.forward(d, out)
torch.cuda.synchronize()
print("synced", flush=True)
got = to_host(f); want = O.ntt(a, T, p, nthreads=4)
bad = np.argwhere(got != want)
print("mismatches", len(bad), "first", bad[:4].tolist())
