"""MI355X (gfx950) constants every roofline figure of this repo is priced on -- ONE place, imported by bench.py and by the
summary / plot tools, so that no two files can state different peaks.

Sources: /opt/skills/guides/MI355X_MICROARCH.md
  * HBM3E ~8 TB/s (spec table)
  * 256 CUs x 4 SIMD-32 units; "a wave (64 lanes) issues each VALU instruction over 2 cycles (32 lanes/cycle x 2)" (Wave
    scheduling); cycle-constants table: `v_fma_f32 (wave64) 2 cyc (SIMD-32); one wave alone: 4`
  * peak engine clock 2.4 GHz
and this repo's own probe of the butterfly streams' instruction forms, profiles/r04_valu_issue_cost.json (tools/valu_issue_cost.hip):
at 8 waves per SIMD the VOP3 forms (SGPR-pair carries, v_cmp_*_u64, v_mad_u64_u32, v_mul_lo_u32, v_lshl_add_u64) issue every
1.91-1.99 cycles and plain VOP1/VOP2 moves / adds every 0.98-1.06 -- the 2-cycle price is the unit's capacity for the forms
these kernels are made of; 4 cycles (rounds 1-4) was what ONE wave alone sustains, not a peak.
"""
HBM_PEAK_GBS = 8000.0
CUS = 256
SIMDS = CUS * 4
PEAK_CLOCK_GHZ = 2.4
VALU_PEAK_CYCLES_PER_WAVE_INSTR = 2.0  # wave64 on a SIMD-32
# the reference's efficiency convention divides "operations" (5.5 N log2 N per transform, profile/plot_efficiency.py:25,44) by a
# peak in GOPS (A100 4280, AIE 88: plot_efficiency.py:27,46); MI355X's 32-bit integer vector peak on the same footing:
# 1024 SIMDs x 32 lanes per clock x 2.4 GHz
PEAK_GOPS_INT32 = SIMDS * 32 * PEAK_CLOCK_GHZ


def valu_frac_of_peak(valu_wave_instr: float, kernel_cycles: float) -> float:
    """VALU wave-instructions of one launch x 2 cycles / (1024 SIMDs x the launch's shader cycles)."""
    return valu_wave_instr * VALU_PEAK_CYCLES_PER_WAVE_INSTR / (SIMDS * kernel_cycles)
