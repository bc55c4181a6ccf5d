#!/bin/bash
# How do the headline kernels respond to FEWER waves per SIMD?  (tools/valu_issue_cost shows the butterfly streams issuing faster at
# 3 waves per SIMD than at 4: 59.0 against 74.7 cycles per butterfly.)  Side build ab/libntt_lds.so = the product sources with ONE
# change: every pass launch asks for NTT_EXTRA_LDS bytes of dynamic LDS, which caps the workgroups per CU (CONTIG 32 KiB static,
# column 36 KiB: +9216 -> 3 per CU, +24576 -> 2).  Same process, interleaved with the product library, per-pass hipEvent times.
set -e
cd "$GRAFT_REPO_ROOT"
for X in 0 9216 24576; do
  echo "## NTT_EXTRA_LDS=$X"
  NTT_EXTRA_LDS=$X python3 tools/ab_pass.py --rounds 5 --reps 5 base=ntt_aie_amd/libntt_hip.so lds=ab/libntt_lds.so 2>&1 | grep -v amdgpu.ids
done
