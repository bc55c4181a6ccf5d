#!/usr/bin/env python3
"""40 inverse transforms of the headline batch (for rocprofv3 --kernel-trace --stats: per-kernel time of the inverse passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from bench import GOLDILOCKS, synth_batch
from ntt_aie_amd import NTTPlan
torch.cuda.set_device(0)
plan = NTTPlan(16, GOLDILOCKS, 8, 0); plan.generate_twiddles(0, 7)
x = synth_batch(torch, 4096, 1 << 16, torch.device("cuda", 0)); y = torch.empty_like(x)
for _ in range(48):
    plan.inverse(x, y)
torch.cuda.synchronize()
