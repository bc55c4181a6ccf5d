import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/tools")
import bench_configs as B
B.run("cfg4: N=2^20, Goldilocks, batch 512, negacyclic polymul", 20, B.GOLD, 7, 8, 512, kind=2, polymul=True)
B.run("N=2^16 polymul batch 4096", 16, B.GOLD, 7, 8, 4096, kind=2, polymul=True)
B.run("N=2^12 polymul batch 65536 (single-pass size)", 12, B.GOLD, 7, 8, 65536, kind=2, polymul=True)
B.run("N=2^8 polymul batch 2^20 (single-pass size)", 8, B.GOLD, 7, 8, 1 << 20, kind=2, polymul=True)
B.run("u32 N=2^12 p=998244353 polymul batch 65536", 12, 998244353, 3, 4, 65536, kind=2, polymul=True)
B.run("u32 N=2^12 p=3221225473 polymul batch 65536", 12, 3221225473, 5, 4, 65536, kind=2, polymul=True)
B.run("u32 N=2^16 p=998244353 polymul batch 8192", 16, 998244353, 3, 4, 8192, kind=2, polymul=True)
B.run("u32 N=2^8 p=998244353 polymul batch 2^20", 8, 998244353, 3, 4, 1 << 20, kind=2, polymul=True)
