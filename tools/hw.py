"""MI355X (gfx950) constants every roofline figure of this repo is priced on -- ONE place, imported by bench.py and by the
summary / plot tools, so that no two files can state different peaks.

HBM3E ~8 TB/s, 256 CUs x 4 SIMDs, peak engine clock 2.4 GHz: /opt/skills/guides/MI355X_MICROARCH.md (spec table).

Vector-ALU throughput per wave64 instruction: MEASURED, profiles/r06_valu_peak.txt (r05's rows reproduced; tools/valu_peak.hip: launches of 12
generations of workgroups, cycles = launch duration x in-kernel clock / wave-instructions per SIMD -- no assumption about how many
waves are resident):
    v_add_co_u32 / v_addc_co_u32 (SGPR-pair carries), v_mad_u64_u32, v_mul_lo_u32, v_cndmask_b32 by SGPR pair   4.04 - 4.08 cycles
    v_mov_b32, v_add_u32                                                                                         2.09 - 2.15 cycles
    v_pk_fma_f32  4.06 cycles = 154 TFLOP/s at 2.4 GHz (the data sheet's 157.3: the method reproduces it)
    v_mul_f32 (VOP2, two operands) 2.09 -- the guide's 2-cycle rate; every FMA form 3.3 - 3.45 (v_fma_f32 with three or two different
    sources, v_fmac_f32): the third operand read costs, whatever the encoding (round 6; MI355X_MICROARCH.md's "v_fma_f32: 2" could not be
    reproduced with this method); the VOP2 carry forms through VCC 4.05 - 4.06, like the SGPR-pair ones
Nominal prices below: 4 cycles (16 lanes per clock) for the VOP3-class forms the butterfly streams are made of, 2 cycles (32 lanes
per clock) for plain VOP1 / VOP2 moves and adds.

History, because two wrong prices were published before this one.  Rounds 1-4 priced every VALU instruction at 4 cycles (from wall-clock
microbenchmarks, profiles/r01_microbench_valu_rates.txt: 1.74 ns) -- right for the streams' forms, 2 x too dear for their two moves.
Round 5 first re-based on 2 cycles per instruction (MI355X_MICROARCH.md: "SIMD-32 ... a wave64 instruction takes 2 cycles"; and
profiles/r04_valu_issue_cost.json showed 1.99 cycles for the VOP3 forms at 8 waves per SIMD).  That probe divides the MEDIAN wave's own
cycle count of a ONE-generation launch by (waves per SIMD x instructions), i.e. it assumes every wave of the launch runs side by side
from start to end; its own wall-clock column says 1.79 ns = 4.3 cycles for the same launches.  tools/valu_peak.hip settles it.
"""
HBM_PEAK_GBS = 8000.0
CUS = 256
SIMDS = CUS * 4
PEAK_CLOCK_GHZ = 2.4
VALU_PEAK_CYCLES_VOP3 = 4.0   # carry forms, 64-bit compare, select by SGPR pair, v_mul_lo/hi_u32, v_mad_u64_u32, v_lshl_add_u64
VALU_PEAK_CYCLES_PLAIN = 2.0  # v_mov_b32, v_add_u32, v_sub_u32, v_and_b32, v_or_b32, v_xor_b32 (VOP1 / VOP2, no carry)
VALU_PEAK_SOURCE = "profiles/r06_valu_peak.txt (tools/valu_peak.hip: steady-state throughput, launch duration x clock / wave-instructions)"
# the reference's efficiency convention divides "operations" (5.5 N log2 N per transform, profile/plot_efficiency.py:25,44) by a
# peak in GOPS (A100 4280, AIE 88: plot_efficiency.py:27,46); MI355X on the same footing: the rate of its FASTEST 32-bit integer
# vector instructions (plain adds: 32 lanes per clock) -- a peak no instruction mix can exceed: 1024 SIMDs x 32 x 2.4 GHz
PEAK_GOPS_INT32 = SIMDS * 32 * PEAK_CLOCK_GHZ


def valu_peak_cycles(valu_instr: float, plain_instr: float = 0.0) -> float:
    """SIMD cycles that `valu_instr` wave-instructions need at the unit's measured throughput, `plain_instr` of them plain moves / adds."""
    return (valu_instr - plain_instr) * VALU_PEAK_CYCLES_VOP3 + plain_instr * VALU_PEAK_CYCLES_PLAIN


def valu_frac_of_peak(valu_wave_instr: float, kernel_cycles: float, plain_share: float = 0.0) -> float:
    """VALU wave-instructions of one launch at their peak price / (1024 SIMDs x the launch's shader cycles).  plain_share = the
    fraction of them that are plain moves / adds (0: every instruction at the VOP3 price -- an UPPER estimate)."""
    return valu_peak_cycles(valu_wave_instr, valu_wave_instr * plain_share) / (SIMDS * kernel_cycles)
