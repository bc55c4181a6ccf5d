import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_configs as B
logn, wb, batch = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
p, g = (B.GOLD, 7) if wb == 8 else (3221225473, 5)
B.run("one", logn, p, g, wb, batch)
