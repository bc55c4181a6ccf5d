#!/bin/bash
# Does the 9-stage column pass pay at other two-pass sizes?  Default split against the split that moves one stage into the column pass,
# 4 GiB of coefficients, same process (tools/ab_latency.py, experiment build + NTT_PLAN_SPLIT).  -> profiles/r03_col9_splits.txt
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
X=ntt_aie_amd/libntt_hip_exp.so
run() { python3 tools/ab_latency.py "$@" 2>&1 | grep -v amdgpu.ids; }
GL="--word-bytes 8 --p 18446744069414584321 --g 7"
for s in "16 4096 8,8 7,9" "17 2048 9,8 8,9" "18 1024 10,8 9,9" "19 512 11,8 10,9" "20 256 12,8 11,9" "21 128 13,8 12,9"; do
  set -- $s
  run --logn $1 $GL --batch $2 --k 10 --rounds 7 d$3=$X+NTT_PLAN_SPLIT=$3 c$4=$X+NTT_PLAN_SPLIT=$4
done
for s in "17 8192 10,7 8,9" "18 4096 10,8 9,9" "19 2048 11,8 10,9" "20 1024 12,8 11,9" "21 512 13,8 12,9"; do
  set -- $s
  run --logn $1 --word-bytes 4 --p 998244353 --g 3 --batch $2 --k 10 --rounds 7 d$3=$X+NTT_PLAN_SPLIT=$3 c$4=$X+NTT_PLAN_SPLIT=$4
done
