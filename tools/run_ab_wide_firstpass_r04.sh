#!/bin/bash
# Does the 512-thread radix-8 variant also pay as the FIRST pass of a multi-pass 4-byte plan at small batches?
# (experiment build: NTT_PASS_VARIANT=1 forces variant 1 on every CONTIG pass of 10..12 stages; 8-byte first passes are radix-8 already)
set -e
cd "$GRAFT_REPO_ROOT"
E=ntt_aie_amd/libntt_hip_exp.so
for N in 16 18 20; do
  for B in 1 8 64 256; do
    for DIR in "" "--inverse"; do
      python3 tools/ab_latency.py --logn $N --p 3221225473 --g 5 --word-bytes 4 --batch $B --rounds 5 --k 30 $DIR r16=$E+NTT_PASS_VARIANT=0 wide=$E+NTT_PASS_VARIANT=1 2>&1 | grep -v amdgpu.ids
    done
  done
done
