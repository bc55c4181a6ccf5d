#!/bin/bash
# NEGATIVE RESULT, kept for provenance (the code it measured was reverted: git show of the commit that adds this file describes it).
# A 64-VGPR build of the 4-byte wide variant -- one kernel per modulus class (pass_kernel<Cfg, SC, MODE>, the class picked on the
# host), the streams' fixed registers at v[56:63] ("_w" twins from tools/gen_gl_asm.py), __launch_bounds__(512, 8): four 512-thread
# workgroups per CU, i.e. config 2's 1024 workgroups in ONE generation at 8 waves per SIMD -- as ntt_aie_amd/libntt_hip.so, against
# the shipped 80-VGPR build (ab/libntt_w80.so: 6 waves, 3 workgroups per CU), a 72-VGPR build (ab/libntt_w72inv.so) and, for the
# crossover, the radix-16 kernel (experiment build, NTT_PASS_VARIANT).  -> profiles/r04_ab_m32_wide64.txt:
# forward -0.3 % at batch 1024 (-3.6 % at 1536), inverse +5 .. +11 % (4 spilled registers): occupancy is not what the launch waits on.
set -e
cd "$GRAFT_REPO_ROOT"
E=ntt_aie_amd/libntt_hip_exp.so
echo "# part 1: product builds, the launcher picks the wide variant (batch below the threshold)"
for CLS in "3221225473 5" "998244353 3"; do
  set -- $CLS
  for N in 12 10; do
    for B in 1 256 512 1024 1536; do
      for DIR in "" "--inverse"; do
        python3 tools/ab_latency.py --logn $N --p $1 --g $2 --word-bytes 4 --batch $B --rounds 7 --k 40 $DIR w80=ab/libntt_w80.so w64=ntt_aie_amd/libntt_hip.so w72=ab/libntt_w72inv.so 2>&1 | grep -v amdgpu.ids
      done
    done
  done
done
echo "# part 2: crossover of the 64-VGPR wide variant against radix-16"
for CLS in "3221225473 5" "2013265921 31" "998244353 3"; do
  set -- $CLS
  for N in 12 11 10; do
    for B in 256 1024 2048 4096 8192; do
      for DIR in "" "--inverse"; do
        python3 tools/ab_latency.py --logn $N --p $1 --g $2 --word-bytes 4 --batch $B --rounds 5 --k 30 $DIR r16=$E+NTT_PASS_VARIANT=0 wide=$E+NTT_PASS_VARIANT=1 2>&1 | grep -v amdgpu.ids
      done
    done
  done
done
