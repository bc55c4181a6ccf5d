#!/bin/bash
# What do the LDS exchanges between the register rounds cost the headline kernels?  Side experiment build ab/libntt_noxchg_exp.so =
# the experiment sources + ONE switch: NTT_DEBUG_FLAGS bit 0x200 skips phase_lds_write / sync / phase_lds_read between rounds (the
# butterflies then run on stale registers: outputs meaningless, VALU work unchanged).  Same process, interleaved, per-pass hipEvents:
# real, no exchange, VALU floor (loads from L2, no stores), VALU floor without exchange.  The upper bound of what ANY exchange
# optimisation (bank-conflict swizzle, fewer barriers) could return.
set -e
cd "$GRAFT_REPO_ROOT"
L=ab/libntt_noxchg_exp.so
python3 tools/ab_pass.py --rounds 5 --reps 5 real=$L+NTT_DEBUG_FLAGS=0 noxchg=$L+NTT_DEBUG_FLAGS=512 floor=$L+NTT_DEBUG_FLAGS=3 floor_noxchg=$L+NTT_DEBUG_FLAGS=515 2>&1 | grep -v amdgpu.ids
