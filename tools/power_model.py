#!/usr/bin/env python3
"""What a saving is worth under the board power cap: a three-term energy model fitted to ONE file, profiles/rNN_power_probe.txt
(tools/power_probe.py: board power and clock while the transform, its VALU work alone, and a copy of its bytes loop).

    P(f) = P_idle + (E_mem + E_valu(f)) / t,      t = C / f,      E_valu(f) = E_valu(f0) * (f / f0)^alpha

  P_idle   the probe's idle line
  E_mem    joules per step of the two HBM trips = 2 x (P_copy - P_idle) x t_copy   (a copy moves one trip's bytes; memory clocks fixed)
  E_valu   joules per step of the butterflies + exchanges at the clock f0 the VALU-only run held = (P_floor - P_idle) x t_floor
  C        shader cycles of one step = t_real x f_real
  alpha    the ONE fitted number: chosen so that the model reproduces the transform's own line (its power at its held clock)

The chip lowers f until P(f) = the power the transform was seen to draw; the model then answers: "C cycles fewer", "E_mem halved"
(a single HBM trip), "fewer / cheaper VALU instructions" -> which clock, which step time.  It is checked against a measurement it
was not fitted to: round 5's 8-waves-per-SIMD upper bound (first pass: kernel cycles -4.8 % -> time -2.9 %).

usage: power_model.py [profiles/rNN_power_probe.txt]  -> text table on stdout.  Pure arithmetic (tests/test_profile_tools.py)."""
import math
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse_probe(text):
    """{name: (ms per iteration, mean W, median sclk GHz)} from tools/power_probe.py's lines."""
    out = {}
    for m in re.finditer(r"^(.+?)\s+([\d.]+) ms/iter\s+power W: max [\d.]+ mean ([\d.]+).*?sclk MHz: \[([\d, ]+)\]", text, re.M):
        cl = sorted(int(x) for x in m.group(4).split(","))
        out[m.group(1).strip()] = (float(m.group(2)), float(m.group(3)), cl[len(cl) // 2] / 1000.0)
    return out


def fit(probe):
    idle = probe["idle-ish (sync only)"][1]
    t_real, p_real, f_real = probe["forward (real)"]
    t_floor, p_floor, f_floor = probe["forward (L2 loads, no stores)"]
    t_copy, p_copy, _ = probe["copy (xor kernel)"]
    e_mem = 2 * (p_copy - idle) * t_copy * 1e-3          # J per step: two trips
    e_valu0 = (p_floor - idle) * t_floor * 1e-3          # J per step at f_floor
    e_dyn_real = (p_real - idle) * t_real * 1e-3         # J per step the transform really spends beyond idle
    e_valu_real = e_dyn_real - e_mem                     # ... of which the butterflies, at f_real
    alpha = math.log(e_valu_real / e_valu0) / math.log(f_real / f_floor)
    return {"P_idle_W": idle, "E_mem_J": e_mem, "E_valu_J_at_f0": e_valu0, "f0_GHz": f_floor, "alpha": alpha,
            "cycles": t_real * 1e-3 * f_real * 1e9, "P_held_W": p_real, "t_real_ms": t_real, "f_real_GHz": f_real,
            "shares_at_operating_point": {"static": idle * t_real * 1e-3, "hbm_trips": e_mem, "butterflies": e_valu_real}}


def solve(m, cycles_scale=1.0, mem_scale=1.0, valu_scale=1.0, f_max=2.4):
    """(clock GHz, step ms) at which the modelled power equals the power the transform was seen to draw; the clock never exceeds f_max."""
    C = m["cycles"] * cycles_scale

    def power(f):
        t = C / (f * 1e9)
        return m["P_idle_W"] + (m["E_mem_J"] * mem_scale + m["E_valu_J_at_f0"] * valu_scale * (f / m["f0_GHz"]) ** m["alpha"]) / t

    if power(f_max) <= m["P_held_W"]:
        return f_max, C / (f_max * 1e9) * 1e3
    lo, hi = 0.5, f_max
    for _ in range(100):
        mid = 0.5 * (lo + hi)
        lo, hi = (mid, hi) if power(mid) <= m["P_held_W"] else (lo, mid)
    return lo, C / (lo * 1e9) * 1e3


def table(m):
    rows = []
    f, t = solve(m)
    rows.append(("the transform as measured (the fit point)", f, t, 0.0))
    for label, kw in (("kernel cycles -5 %", dict(cycles_scale=0.95)), ("kernel cycles -10 %", dict(cycles_scale=0.90)),
                      ("kernel cycles -20 %", dict(cycles_scale=0.80)),
                      ("one HBM trip instead of two (E_mem halved), same cycles", dict(mem_scale=0.5)),
                      ("one HBM trip AND kernel cycles -10 %", dict(mem_scale=0.5, cycles_scale=0.90)),
                      ("butterfly energy -10 % (fewer / cheaper VALU instructions), cycles -5 %", dict(valu_scale=0.90, cycles_scale=0.95)),
                      ("first pass only: its kernel cycles -4.8 % (round 5's 8-wave upper bound) = step cycles -2.4 %", dict(cycles_scale=0.976))):
        f2, t2 = solve(m, **kw)
        rows.append((label, f2, t2, 100 * (t2 / t - 1)))
    return rows


def main():
    import glob

    path = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_power_probe.txt")))[-1]
    m = fit(parse_probe(open(path).read()))
    sh = m["shares_at_operating_point"]
    tot = sum(sh.values())
    print("# tools/power_model.py on %s" % os.path.relpath(path, ROOT))
    print("# P_idle %.0f W; E_mem %.3f J per step (two trips); E_valu %.3f J per step at %.2f GHz; alpha %.2f (fitted to the transform's own line: "
          "%.0f W at %.2f GHz, %.4f ms)" % (m["P_idle_W"], m["E_mem_J"], m["E_valu_J_at_f0"], m["f0_GHz"], m["alpha"], m["P_held_W"], m["f_real_GHz"], m["t_real_ms"]))
    print("# joules per step at the operating point: static %.2f (%.0f %%), HBM trips %.2f (%.0f %%), butterflies + exchanges %.2f (%.0f %%)"
          % (sh["static"], 100 * sh["static"] / tot, sh["hbm_trips"], 100 * sh["hbm_trips"] / tot, sh["butterflies"], 100 * sh["butterflies"] / tot))
    print("%-100s %9s %9s %8s" % ("what changes", "clock GHz", "step ms", "time"))
    for label, f, t, d in table(m):
        print("%-100s %9.2f %9.3f %+7.1f%%" % (label, f, t, d))
    print("# check against a measurement the model was not fitted to (profiles/r05_ab_contig8w_bound.txt): first-pass kernel cycles -4.8 % at the\n"
          "# same held clock, step 1.6384 -> 1.6130 ms = -1.55 % measured; the last row is the model's figure for the same change.")


if __name__ == "__main__":
    main()
