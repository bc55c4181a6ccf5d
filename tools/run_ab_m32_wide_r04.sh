#!/bin/bash
# Crossover of the 4-byte "wide" CONTIG variant (PassDesc::variant 1: 512 threads x 8 words, radix-8 rounds) against the default
# radix-16 kernel (256 threads x 16 words) at N = 2^10 .. 2^12: same process, interleaved, outputs compared word for word.
# Experiment build: NTT_PASS_VARIANT forces the variant of every CONTIG pass.  -> profiles/r04_ab_m32_wide.txt
set -e
cd "$GRAFT_REPO_ROOT"
E=ntt_aie_amd/libntt_hip_exp.so
for CLS in "3221225473 5" "2013265921 31" "998244353 3"; do
  set -- $CLS
  for N in 12 11 10; do
    for B in 1 64 256 1024 4096 16384 65536; do
      for DIR in "" "--inverse"; do
        python3 tools/ab_latency.py --logn $N --p $1 --g $2 --word-bytes 4 --batch $B --rounds 5 --k 30 $DIR r16=$E+NTT_PASS_VARIANT=0 wide=$E+NTT_PASS_VARIANT=1 2>&1 | grep -v amdgpu.ids
      done
    done
  done
done
