V="lin=ab/libntt_lin.so xwl=ab/libntt_xwl.so"
python tools/ab_pass.py --logn 20 --batch 512 --rounds 5 --reps 3 $V
python tools/ab_pass.py --logn 18 --batch 2048 --rounds 5 --reps 3 $V
python tools/ab_pass.py --logn 16 --batch 4096 --rounds 5 --reps 3 $V
python tools/ab_latency.py $V
python tools/ab_latency.py --inverse $V
python tools/ab_latency.py --batch 65536 --k 10 $V
python tools/ab_latency.py --p 998244353 --g 3 --batch 65536 --k 10 $V
python tools/ab_latency.py --logn 12 --p 18446744069414584321 --g 7 --word-bytes 8 --batch 16384 --k 10 $V
for v in lin xwl; do echo "== polymul $v"; NTT_HIP_LIB=$PWD/ab/libntt_$v.so python tools/bench_polymul.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'], 'polymul_ms', round(d.get('polymul_ms', 0), 4), 'fwd', round(d['forward_ms'], 4), 'inv', round(d['inverse_ms'], 4))
"; done
