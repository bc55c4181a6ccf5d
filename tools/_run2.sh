V="new=ab/libntt_new.so early=ab/libntt_early.so"
python tools/ab_latency.py --inverse $V
python tools/ab_latency.py --inverse --batch 65536 --k 10 $V
python tools/ab_latency.py --inverse --p 998244353 --g 3 --batch 65536 --k 10 $V
python tools/ab_latency.py --inverse --logn 8 --p 3329 --g 3 --batch 1048576 --k 10 $V
python tools/ab_latency.py --logn 16 --p 998244353 --g 3 --batch 8192 --k 10 $V
python tools/ab_latency.py --inverse --logn 16 --p 998244353 --g 3 --batch 8192 --k 10 $V
python tools/ab_latency.py --logn 16 --p 3221225473 --g 5 --batch 64 --k 50 $V
python tools/ab_latency.py $V
