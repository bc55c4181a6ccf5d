set -e
cd "$GRAFT_REPO_ROOT"
E=ntt_aie_amd/libntt_hip_exp.so
for N in 12 11 10; do
  for B in 1 64 256 1024 4096 16384; do
    for DIR in "" "--inverse"; do
      python3 tools/ab_latency.py --logn $N --p 18446744069414584321 --g 7 --word-bytes 8 --batch $B --rounds 5 --k 30 $DIR r16=$E+NTT_PASS_VARIANT=0 wide=$E+NTT_PASS_VARIANT=1 2>&1 | grep -v amdgpu.ids
    done
  done
done
