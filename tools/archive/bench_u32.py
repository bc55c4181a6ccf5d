#!/usr/bin/env python3
"""4-byte-word throughput at the BASELINE shapes (A/B with NTT_HIP_LIB=ab/libntt_NAME.so): forward / inverse ms."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_configs as B
B.run("cfg2: N=2^12, p=3221225473 (any), batch 1024", 12, 3221225473, 5, 4, 1024)
B.run("cfg2c: N=2^12, p=3221225473 (any), batch 65536", 12, 3221225473, 5, 4, 65536)
B.run("N=2^12, p=2013265921 (small: 2^30 <= p < 2^31), batch 65536", 12, 2013265921, 31, 4, 65536)
B.run("cfg2d: N=2^12, p=998244353 (lazy), batch 65536", 12, 998244353, 3, 4, 65536)
B.run("kyber-like: N=2^8, p=3329, batch 2^20", 8, 3329, 3, 4, 1 << 20)
B.run("N=2^16, p=998244353 (lazy), batch 8192", 16, 998244353, 3, 4, 8192)
