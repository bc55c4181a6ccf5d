cd "$GRAFT_REPO_ROOT"
X=ntt_aie_amd/libntt_hip_exp.so
run() { python3 tools/ab_latency.py "$@" 2>&1 | grep -v amdgpu.ids; }
for f in "--word-bytes 8 --p 18446744069414584321 --g 7" "--word-bytes 4 --p 998244353 --g 3" "--word-bytes 4 --p 3221225473 --g 5"; do
  for b in 8 16 32 64; do
    run --logn 22 $f --batch $b --k 20 --rounds 7 three=$X+NTT_PLAN_SPLIT=8,7,7 two13_9=$X+NTT_PLAN_SPLIT=13,9
    run --logn 22 $f --batch $b --k 20 --rounds 7 --inverse three=$X+NTT_PLAN_SPLIT=8,7,7 two13_9=$X+NTT_PLAN_SPLIT=13,9
  done
done
