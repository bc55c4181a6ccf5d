#!/bin/bash
# N = 2^22: 8 + 7 + 7 (three passes) against 13 + 9 (two passes, the 512-row column tile), by word size / modulus class and batch.
# -> profiles/r03_n22_classes.txt (the crossovers behind plan.h: plan_alternatives for n == 22)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
X=ntt_aie_amd/libntt_hip_exp.so
run() { python3 tools/ab_latency.py "$@" 2>&1 | grep -v amdgpu.ids; }
for f in "--word-bytes 8 --p 18446744069414584321 --g 7" "--word-bytes 4 --p 998244353 --g 3" "--word-bytes 4 --p 2013265921 --g 31" "--word-bytes 4 --p 3221225473 --g 5"; do
  for b in 1 2 4 8 16 32 64 128 256; do
    [ "$b" = 256 ] && [[ "$f" == *"bytes 8"* ]] && continue
    k=$([ $b -gt 16 ] && echo 3 || echo 15)
    run --logn 22 $f --batch $b --k $k --rounds 5 three=$X+NTT_PLAN_SPLIT=8,7,7 two13_9=$X+NTT_PLAN_SPLIT=13,9
  done
  for b in 1 16 128; do
    run --logn 22 $f --batch $b --k 3 --rounds 5 --inverse three=$X+NTT_PLAN_SPLIT=8,7,7 two13_9=$X+NTT_PLAN_SPLIT=13,9
  done
done
