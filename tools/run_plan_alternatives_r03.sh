#!/bin/bash
# Crossovers behind plan.h: plan_alternatives (round 3).  Same process, interleaved (tools/ab_latency.py), the experiment build
# with NTT_PLAN_SPLIT pinning each decomposition; outputs compared word for word.  -> profiles/r03_plan_alternatives.txt
set -eo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
X=ntt_aie_amd/libntt_hip_exp.so
GL="--word-bytes 8 --p 18446744069414584321 --g 7"
run() { python3 tools/ab_latency.py "$@" 2>&1 | grep -v amdgpu.ids; }
echo "# 8-byte N = 2^13: 7 + 6 against one 13-stage pass, by batch"
for b in 1 32 64 128 256 512 1024 2048 4096; do run --logn 13 $GL --batch $b --k 60 --rounds 5 two=$X+NTT_PLAN_SPLIT=7,6 one13=$X+NTT_PLAN_SPLIT=13; done
run --logn 13 $GL --batch 65536 --k 5 --rounds 5 two=$X+NTT_PLAN_SPLIT=7,6 one13=$X+NTT_PLAN_SPLIT=13
for b in 256 512 1024 65536; do run --logn 13 $GL --batch $b --k $([ $b -gt 4096 ] && echo 5 || echo 60) --rounds 5 --inverse two=$X+NTT_PLAN_SPLIT=7,6 one13=$X+NTT_PLAN_SPLIT=13; done
echo "# 4-byte lazy prime (998244353) N = 2^14: 8 + 6 against one 14-stage pass, by batch"
M="--word-bytes 4 --p 998244353 --g 3"
for b in 1 32 64 128 256 512 1024 4096; do run --logn 14 $M --batch $b --k 60 --rounds 5 two=$X+NTT_PLAN_SPLIT=8,6 one14=$X+NTT_PLAN_SPLIT=14; done
run --logn 14 $M --batch 65536 --k 5 --rounds 5 two=$X+NTT_PLAN_SPLIT=8,6 one14=$X+NTT_PLAN_SPLIT=14
for b in 256 1024 65536; do run --logn 14 $M --batch $b --k $([ $b -gt 4096 ] && echo 5 || echo 60) --rounds 5 --inverse two=$X+NTT_PLAN_SPLIT=8,6 one14=$X+NTT_PLAN_SPLIT=14; done
echo "# 4-byte 32-bit prime N = 2^14 (the 14-stage pass is NOT offered for this class)"
run --logn 14 --word-bytes 4 --p 3221225473 --g 5 --batch 65536 --k 5 --rounds 5 two=$X+NTT_PLAN_SPLIT=8,6 one14=$X+NTT_PLAN_SPLIT=14
echo "# N = 2^22: 8 + 7 + 7 (round 2) against 13 + 9 (the 9-stage column pass)"
for b in 1 8 128; do run --logn 22 $GL --batch $b --k $([ $b -gt 8 ] && echo 3 || echo 20) --rounds 5 three=$X+NTT_PLAN_SPLIT=8,7,7 two13_9=$X+NTT_PLAN_SPLIT=13,9; done
run --logn 22 $GL --batch 128 --k 3 --rounds 5 --inverse three=$X+NTT_PLAN_SPLIT=8,7,7 two13_9=$X+NTT_PLAN_SPLIT=13,9
for pg in "998244353 --g 3" "3221225473 --g 5"; do
  for b in 1 256; do run --logn 22 --word-bytes 4 --p $pg --batch $b --k $([ $b -gt 8 ] && echo 3 || echo 20) --rounds 5 three=$X+NTT_PLAN_SPLIT=8,7,7 two13_9=$X+NTT_PLAN_SPLIT=13,9; done
done
run --logn 22 $M --batch 256 --k 3 --rounds 5 two13_9=$X+NTT_PLAN_SPLIT=13,9 two14_8=$X+NTT_PLAN_SPLIT=14,8
echo "# 4-byte N = 2^16 by modulus class: 8 + 8 against 10 + 6"
for pg in "998244353 --g 3" "2013265921 --g 31" "3221225473 --g 5"; do
  run --logn 16 --word-bytes 4 --p $pg --batch 16384 --k 5 --rounds 7 s8_8=$X+NTT_PLAN_SPLIT=8,8 s10_6=$X+NTT_PLAN_SPLIT=10,6
  run --logn 16 --word-bytes 4 --p $pg --batch 1 --k 60 --rounds 5 s8_8=$X+NTT_PLAN_SPLIT=8,8 s10_6=$X+NTT_PLAN_SPLIT=10,6
done
