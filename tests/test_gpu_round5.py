"""Round 5 on the GPU: the price list the rooflines are computed on is a MEASUREMENT of this box, not a constant from a document.
tools/hw.py says 4 cycles per wave64 instruction for the VOP3-class forms of the butterfly statements and 2 for plain moves / adds;
tools/valu_peak (built by __graft_entry__.build()) measures them -- steady state, launch duration x in-kernel clock / wave-instructions
per SIMD -- and its v_pk_fma_f32 row must reproduce the chip's published FP32 vector peak, which pins the method itself."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_valu_price_list_is_what_this_gpu_measures():
    import hw

    exe = os.path.join(ROOT, "tools", "valu_peak")
    if not os.path.exists(exe):
        pytest.skip("tools/valu_peak not built (a measurement tool: __graft_entry__.build() builds it best-effort)")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    rows = {}
    for line in out.stdout.splitlines():
        if line.startswith("v_"):  # a 72-character label, then: ns, clock GHz, cycles[, TFLOP/s]
            nums = line[72:].split()
            rows[line.split()[0]] = {"ns": float(nums[0]), "ghz": float(nums[1]), "cycles": float(nums[2])}
    for name in ("v_add_co_u32", "v_addc_co_u32", "v_mad_u64_u32", "v_mul_lo_u32", "v_cndmask_b32"):
        assert 0.93 * hw.VALU_PEAK_CYCLES_VOP3 < rows[name]["cycles"] < 1.10 * hw.VALU_PEAK_CYCLES_VOP3, (name, rows[name])
    for name in ("v_mov_b32", "v_add_u32"):
        assert 0.93 * hw.VALU_PEAK_CYCLES_PLAIN < rows[name]["cycles"] < 1.20 * hw.VALU_PEAK_CYCLES_PLAIN, (name, rows[name])
    # the method against the data sheet: v_pk_fma_f32 = 4 FLOP per lane per instruction; at the peak clock that is the published 157.3 TFLOP/s
    pk = rows["v_pk_fma_f32"]
    tflops_at_peak_clock = 4 * 64 / pk["cycles"] * hw.PEAK_CLOCK_GHZ * 1e9 * hw.SIMDS / 1e12
    assert 0.92 * 157.3 < tflops_at_peak_clock < 1.05 * 157.3, (pk, tflops_at_peak_clock)
    assert 1.0 < rows["v_mov_b32"]["ghz"] < 2.6  # the in-kernel clock the cycles are computed with is a sane shader clock
    # round 6: the rows that were to reconcile v_fma_f32's 3.3 cycles with the guide's 2.  They do not: the VOP2
    # two-source forms (v_fmac_f32, v_mul_f32) cost what the three-source VOP3 form costs, so the 3.3 is not an operand-port effect of
    # the probe -- the guide's 2-cycle row is not reproducible with this method on this part.  Asserted here: the
    # f32 forms sit together between the plain moves (2) and the VOP3-class integer forms (4), and the VOP2 carry forms are priced
    # like the SGPR-pair ones (the statements' prices do not depend on which carry register a form names).
    for name in ("v_fma_f32", "v_fmac_f32", "v_fma_f32_2src", "v_mul_f32"):
        assert name in rows and 2.0 < rows[name]["cycles"] < 4.3, (name, rows.get(name))
    f32 = [rows[k]["cycles"] for k in ("v_fma_f32", "v_fmac_f32", "v_fma_f32_2src")]
    assert max(f32) / min(f32) < 1.35, f32
    for name in ("v_add_co_u32_e32", "v_addc_co_u32_e32"):
        assert name in rows and 1.8 < rows[name]["cycles"] < 1.10 * hw.VALU_PEAK_CYCLES_VOP3, (name, rows.get(name))
