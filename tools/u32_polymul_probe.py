import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from ntt_aie_amd import NTTPlan
pl = NTTPlan(12, 998244353, 4, 0); pl.set_twiddles(pl.make_table(2, 3))
a = torch.randint(0, 998244353, (65536, 4096), dtype=torch.int64, device="cuda:0").to(torch.int32); b = a.clone()
for _ in range(5): pl.polymul_negacyclic(a, b)
torch.cuda.synchronize()
