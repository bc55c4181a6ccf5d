"""The counter summaries that bench.py quotes (tools/pmc_summary.py, tools/sq_summary.py) key a kernel by its FULL
PassCfg<...> argument list: a CSV holding the forward AND the inverse kernel of one pass shape yields two entries, and
bench.py's forward_counters() can only ever return the forward one.  CPU only: synthetic CSVs."""
import csv
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))

HDR = ["Correlation_Id", "Dispatch_Id", "Agent_Id", "Queue_Id", "Process_Id", "Thread_Id", "Grid_Size", "Kernel_Id", "Kernel_Name",
       "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name",
       "Counter_Value", "Start_Timestamp", "End_Timestamp"]


def kname(log_m, log_c, contig, inv, log_e):
    cfg = "ntt::PassCfg<ntt::FieldGL, %d, %d, %s, %s, 15, %d, 8, true>" % (log_m, log_c, str(contig).lower(), str(inv).lower(), log_e)
    return "void ntt::(anonymous namespace)::pass_kernel<%s >(ntt::PassArgs<%s >)" % (cfg, cfg)


KERNELS = {  # (contig, inv) -> name
    (True, False): kname(8, 0, True, False, 3), (True, True): kname(8, 0, True, True, 3),
    (False, False): kname(8, 4, False, False, 4), (False, True): kname(8, 4, False, True, 4),
}
OTHER = "void at::native::vectorized_elementwise_kernel<4, at::native::BinaryFunctor<long> >(int)"


def write_csv(path, rows):
    with open(path, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(HDR)
        for i, (name, grid, counter, value, t0, t1) in enumerate(rows):
            w.writerow([i, i, "Agent 2", 1, 1, 1, grid, 7, name, 256, 32768, 0, 96, 0, 96, counter, value, t0, t1])


def test_parse_pass_kernel_full_argument_list():
    from kernel_key import parse_pass_kernel

    f, i = parse_pass_kernel(KERNELS[(True, False)]), parse_pass_kernel(KERNELS[(True, True)])
    assert f["key"] != i["key"] and not f["inv"] and i["inv"]
    assert f["short"] == "pass_contig_8_fwd" and i["short"] == "pass_contig_8_inv"
    assert f["key"] == "PassCfg<ntt::FieldGL, 8, 0, true, false, 15, 3, 8, true>"
    c = parse_pass_kernel(KERNELS[(False, True)])
    assert c["short"] == "pass_col_8_inv" and c["log_c"] == 4 and c["log_e"] == 4 and c["field"] == "FieldGL"
    assert parse_pass_kernel(OTHER) is None


def kname_sc(log_m, log_c, contig, inv, log_e, sc):
    """Round 3's kernel names: pass_kernel<Cfg, SC>."""
    cfg = "ntt::PassCfg<ntt::FieldGL, %d, %d, %s, %s, 15, %d, 8, true>" % (log_m, log_c, str(contig).lower(), str(inv).lower(), log_e)
    return "void ntt::(anonymous namespace)::pass_kernel<%s, %s>(ntt::PassArgs<%s >)" % (cfg, str(sc).lower(), cfg)


def test_scaled_and_unscaled_inverse_kernels_are_two_keys(tmp_path):
    """pass_kernel<Cfg, true> (N^-1 folded into stage 0) and pass_kernel<Cfg, false> of ONE PassCfg are different kernels:
    two keys, two short names, never averaged (sq_summary) nor overwritten (pmc_summary); a stored key parses back."""
    import pmc_summary
    import sq_summary
    from kernel_key import forward_entry, parse_pass_kernel

    plain, scaled = kname_sc(8, 0, True, True, 3, False), kname_sc(8, 0, True, True, 3, True)
    a, b = parse_pass_kernel(plain), parse_pass_kernel(scaled)
    assert not a["sc"] and b["sc"] and a["cfg"] == b["cfg"] and a["key"] != b["key"]
    assert a["short"] == "pass_contig_8_inv" and b["short"] == "pass_contig_8_inv_sc"
    assert a["key"] == "PassCfg<ntt::FieldGL, 8, 0, true, true, 15, 3, 8, true>" and b["key"] == a["key"] + " +SC"
    back = parse_pass_kernel("pass_kernel<" + b["key"])  # how forward_entry() re-parses a stored key
    assert back["sc"] and back["key"] == b["key"] and back["inv"]
    fwd = kname_sc(8, 0, True, False, 3, False)
    assert parse_pass_kernel(fwd)["key"] == "PassCfg<ntt::FieldGL, 8, 0, true, false, 15, 3, 8, true>"  # forward keys unchanged
    wave_bf = 4096 * 32768 * 8 / 64
    rows = []
    for name, ipb in ((plain, 21.0), (scaled, 24.0), (fwd, 23.0)):
        for rep in range(2):
            rows.append((name, 2097152, "SQ_INSTS_VALU", ipb * wave_bf, 1000, 801000))
            rows.append((name, 2097152, "GRBM_GUI_ACTIVE", 8 * 1.6e6, 1000, 801000))
    p = tmp_path / "sq.csv"
    write_csv(p, rows)
    d = sq_summary.summarize(str(p), batch=4096, logn=16, src_hash="abc")
    by_short = {v["short"]: v for v in d["kernels"].values()}
    assert set(by_short) == {"pass_contig_8_inv", "pass_contig_8_inv_sc", "pass_contig_8_fwd"}
    assert by_short["pass_contig_8_inv"]["valu_instr_per_butterfly"] == pytest.approx(21.0)
    assert by_short["pass_contig_8_inv_sc"]["valu_instr_per_butterfly"] == pytest.approx(24.0)
    hit, why = forward_entry(d["kernels"], True, 8)
    assert why is None and hit[1]["short"] == "pass_contig_8_fwd"
    kib = 4 * 1024 * 1024
    fr = [(n, 2097152, "FETCH_SIZE", kib / 4 + x, 1, 2) for n, x in ((plain, 0), (scaled, 512))]
    wr = [(n, 2097152, "WRITE_SIZE", kib / 2 + x, 1, 2) for n, x in ((plain, 0), (scaled, 512))]
    pf, pw = tmp_path / "f.csv", tmp_path / "w.csv"
    write_csv(pf, fr)
    write_csv(pw, wr)
    t = pmc_summary.summarize(str(pf), str(pw), src_hash="abc")
    assert len(t["kernels"]) == 2 and len({v["hbm_bytes_per_launch"] for v in t["kernels"].values()}) == 2


def test_sq_summary_keeps_directions_apart(tmp_path):
    import sq_summary

    wave_bf = 4096 * 32768 * 8 / 64
    rows = []
    # forward kernels: 22.9 / 22.3 instructions per butterfly; inverse: 25 / 21 -- round 2's summary averaged them
    for (contig, inv), ipb, cyc in (((True, False), 22.9, 1.6e6), ((True, True), 25.0, 1.8e6), ((False, False), 22.3, 1.55e6),
                                    ((False, True), 21.0, 1.5e6)):
        for rep in range(3):
            t0 = 1000000 * (rep + 1)
            rows.append((KERNELS[(contig, inv)], 2097152, "SQ_INSTS_VALU", ipb * wave_bf, t0, t0 + 800000))
            rows.append((KERNELS[(contig, inv)], 2097152, "SQ_ACTIVE_INST_VALU", ipb * wave_bf, t0, t0 + 800000))
            rows.append((KERNELS[(contig, inv)], 2097152, "GRBM_GUI_ACTIVE", 8 * cyc, t0, t0 + 800000))
            rows.append((KERNELS[(contig, inv)], 2097152, "SQ_WAVE_CYCLES", 1.0e9, t0, t0 + 800000))
        rows.append((KERNELS[(contig, inv)], 4096, "SQ_INSTS_VALU", 1.0, 1, 2))  # a tiny warm-up launch: filtered by grid size
    rows.append((OTHER, 4194304, "SQ_INSTS_VALU", 1e9, 1, 2))
    p = tmp_path / "sq.csv"
    write_csv(p, rows)
    d = sq_summary.summarize(str(p), batch=4096, logn=16, src_hash="abc")
    assert len(d["kernels"]) == 4  # two shapes x two directions = four keys
    by_short = {v["short"]: v for v in d["kernels"].values()}
    assert set(by_short) == {"pass_contig_8_fwd", "pass_contig_8_inv", "pass_col_8_fwd", "pass_col_8_inv"}
    assert by_short["pass_contig_8_fwd"]["valu_instr_per_butterfly"] == pytest.approx(22.9)
    assert by_short["pass_contig_8_inv"]["valu_instr_per_butterfly"] == pytest.approx(25.0)
    assert by_short["pass_col_8_fwd"]["valu_instr_per_butterfly"] == pytest.approx(22.3)
    assert by_short["pass_contig_8_fwd"]["launches"] == 3
    assert by_short["pass_contig_8_fwd"]["held_clock_GHz"] == pytest.approx(1.6e6 / 800000.0)  # cycles per ns
    assert by_short["pass_contig_8_fwd"]["direction"] == "fwd" and by_short["pass_col_8_inv"]["direction"] == "inv"

    # bench.py picks entries through kernel_key.forward_entry: never an inverse kernel
    import bench

    passes = [("contig", 0, 8), ("col", 8, 8)]
    ent, why = bench.forward_counters(d, passes)
    assert why is None and [e[1]["direction"] for e in ent] == ["fwd", "fwd"]
    assert [e[1]["valu_instr_per_butterfly"] for e in ent] == [pytest.approx(22.9), pytest.approx(22.3)]
    v = bench.valu_roofline(ent, passes, [0.83, 0.82], 4096, 16)
    assert v["instr_per_butterfly"] == [pytest.approx(22.9), pytest.approx(22.3)]
    assert v["peak_butterflies_per_s"] == pytest.approx(1024 * 2.4e9 / (4 * 22.6) * 64)  # no mix given: every instruction at the VOP3-class price of 4 cycles
    assert 0 < v["frac_of_peak_at_2.4GHz"] < 1 and 0 < v["frac_of_peak_at_held_clock"] <= 1
    # the summary's own per-kernel figure is the same quotient (all instructions at 4 cycles: the upper estimate)
    assert by_short["pass_contig_8_fwd"]["valu_frac_of_peak_all_vop3"] == pytest.approx(22.9 * wave_bf * 4 / (1024 * 1.6e6))
    assert v["frac_of_peak_at_held_clock_per_pass"][0] == pytest.approx(by_short["pass_contig_8_fwd"]["valu_frac_of_peak_all_vop3"])
    # a summary holding ONLY inverse kernels (what round 2's traffic file was) yields nothing
    only_inv = {"kernels": {k: e for k, e in d["kernels"].items() if e["direction"] == "inv"}}
    ent, why = bench.forward_counters(only_inv, passes)
    assert ent is None and "no forward" in why


def test_pmc_summary_keeps_directions_apart(tmp_path):
    import pmc_summary

    kib = 4 * 1024 * 1024  # 4 GiB in KiB
    fr, wr = [], []
    for (contig, inv), extra in (((True, False), 0), ((True, True), 1000), ((False, False), 0), ((False, True), 2000)):
        for rep in range(2):
            fr.append((KERNELS[(contig, inv)], 2097152, "FETCH_SIZE", kib / 4 + extra, 1, 2))  # the counter sees half of 2 GiB read
            wr.append((KERNELS[(contig, inv)], 2097152, "WRITE_SIZE", kib / 2 + extra, 1, 2))
        fr.append((KERNELS[(contig, inv)], 4096, "FETCH_SIZE", 16.0, 1, 2))  # parity-sized launch: dropped (< half of the largest)
        wr.append((KERNELS[(contig, inv)], 4096, "WRITE_SIZE", 16.0, 1, 2))
    pf, pw = tmp_path / "f.csv", tmp_path / "w.csv"
    write_csv(pf, fr)
    write_csv(pw, wr)
    d = pmc_summary.summarize(str(pf), str(pw), src_hash="abc")
    assert len(d["kernels"]) == 4
    fwd = [v for v in d["kernels"].values() if v["direction"] == "fwd"]
    assert len(fwd) == 2 and all(v["hbm_bytes_per_launch"] == pytest.approx(2 ** 32) for v in fwd)
    inv = [v for v in d["kernels"].values() if v["direction"] == "inv"]
    assert all(v["hbm_bytes_per_launch"] > 2 ** 32 for v in inv) and all(v["launches"] == 2 for v in inv)
    import bench

    ent, why = bench.forward_counters(d, [("contig", 0, 8), ("col", 8, 8)])
    assert why is None and sum(e[1]["hbm_bytes_per_launch"] for e in ent) == pytest.approx(2 ** 33)


PRODUCT = ("void ntt::(anonymous namespace)::product_kernel<ntt::PassCfg<ntt::FieldGL, 12, 0, true, true, 9, 3, 9, false>, "
           "ntt::PassCfg<ntt::FieldGL, 12, 0, true, false, 9, 3, 9, false> >(ntt::PassArgs<ntt::PassCfg<ntt::FieldGL, 12, 0, true, true, 9, 3, 9, false> >, "
           "ntt::PassCfg<ntt::FieldGL, 12, 0, true, true, 9, 3, 9, false>::W const*, ntt::PassArgs<ntt::PassCfg<ntt::FieldGL, 12, 0, true, false, 9, 3, 9, false> >)")


def test_config_summary_per_operation(tmp_path):
    """tools/config_summary.py (configs 2 and 4 under the headline's evidence standard): the product's kernels -- fused middle
    pass, inverse column pass twice per operation (once per operand), forward column pass once -- reduce to HBM bytes and VALU
    instructions PER OPERATION; the product kernel is recognised (kernel_key.parse_kernel) and charged three 12-stage networks."""
    import config_summary as CS
    from configs import CONFIGS, algorithmic_bytes, butterflies
    from kernel_key import parse_kernel

    pk = parse_kernel(PRODUCT)
    assert pk["kind"] == "product" and pk["log_m"] == 12 and pk["short"] == "product_12" and pk["stage_legs"] == 3
    assert parse_kernel(kname_sc(8, 4, False, True, 4, False))["kind"] == "pass" and parse_kernel(OTHER) is None
    c = CONFIGS["cfg4"]
    n, batch = 1 << 20, 512
    assert algorithmic_bytes(c) == 9 * n * 8 * batch and butterflies(c) == 3 * batch * (n // 2) * 20
    inv_col, fwd_col = kname_sc(8, 12, False, True, 4, False), kname_sc(8, 12, False, False, 4, False)
    ops = 4
    word_kib = n * 8 * batch / 1024.0  # one pass over one operand, in KiB
    fr, wr, sq = [], [], []
    for op in range(ops):
        for name, launches, rd, wrt, ipb, stages, legs in ((inv_col, 2, 1, 1, 21.4, 8, 1), (PRODUCT, 1, 2, 1, 24.0, 12, 3), (fwd_col, 1, 1, 1, 22.2, 8, 1)):
            for l in range(launches):
                t0 = 1000 * (op * 10 + l)
                fr.append((name, 1 << 24, "FETCH_SIZE", rd * word_kib / 2, t0, t0 + 900))   # the counter sees half of the bytes read
                wr.append((name, 1 << 24, "WRITE_SIZE", wrt * word_kib, t0, t0 + 900))
                wave_bf = legs * batch * (n // 2) * stages / 64.0
                sq.append((name, 1 << 24, "SQ_INSTS_VALU", ipb * wave_bf, t0, t0 + 900))
                sq.append((name, 1 << 24, "GRBM_GUI_ACTIVE", 8 * 3.0e6, t0, t0 + 900))
                sq.append((name, 1 << 24, "SQ_WAVE_CYCLES", 3.0e9, t0, t0 + 900))
    fr.append((OTHER, 1 << 24, "FETCH_SIZE", 1e9, 1, 2))
    pf, pw, ps = tmp_path / "f.csv", tmp_path / "w.csv", tmp_path / "s.csv"
    write_csv(pf, fr)
    write_csv(pw, wr)
    write_csv(ps, sq)
    t = CS.summarize_pmc("cfg4", ops, str(pf), str(pw), src_hash="abc")
    assert len(t["kernels"]) == 3 and t["src_hash"] == "abc" and t["config"] == "cfg4"
    by = {v["short"]: v for v in t["kernels"].values()}
    assert by["pass_col_8_inv"]["launches_per_op"] == 2 and by["product_12"]["launches_per_op"] == 1
    # inverse column passes 2 x 2N, fused middle 3N, forward column 2N = the 9N words the product is priced on
    assert t["per_op"]["hbm_bytes"] == pytest.approx(9 * n * 8 * batch) and t["per_op"]["ratio_to_algorithmic"] == pytest.approx(1.0)
    s = CS.summarize_sq("cfg4", ops, str(ps), src_hash="abc")
    bs = {v["short"]: v for v in s["kernels"].values()}
    assert bs["pass_col_8_inv"]["valu_instr_per_butterfly"] == pytest.approx(21.4)
    assert bs["product_12"]["valu_instr_per_butterfly"] == pytest.approx(24.0)
    assert bs["pass_col_8_fwd"]["valu_instr_per_butterfly"] == pytest.approx(22.2)
    assert bs["product_12"]["kernel_cycles"] == pytest.approx(3.0e6) and bs["product_12"]["held_clock_GHz"] == pytest.approx(3.0e6 / 900)
    # per operation: 60 stage-networks of N/2 butterflies per polynomial... 2x8 inverse + 3x12 middle + 8 forward = 60 = 3 x 20 stages
    assert s["per_op"]["butterflies_in_profiled_kernels"] == pytest.approx(butterflies(c))
    want = (21.4 * 16 + 24.0 * 36 + 22.2 * 8) / 60
    assert s["per_op"]["valu_instr_per_butterfly"] == pytest.approx(want)
    assert s["per_op"]["kernel_cycles"] == pytest.approx(4 * 3.0e6)  # four launches per operation
    # a single-pass forward configuration: one kernel, one launch per operation
    c2 = CONFIGS["cfg2"]
    k2 = "void ntt::(anonymous namespace)::pass_kernel<ntt::PassCfg<ntt::FieldM32, 12, 0, true, false, 15, 4, 8, true>, false>(x)"
    rows = [(k2, 1 << 18, "SQ_INSTS_VALU", 12.0 * c2["batch"] * 2048 * 12 / 64, 0, 14000), (k2, 1 << 18, "GRBM_GUI_ACTIVE", 8 * 30000.0, 0, 14000),
            (k2, 256, "SQ_INSTS_VALU", 5.0, 0, 10)]  # ... and a parity-sized launch of the same kernel: dropped
    write_csv(ps, rows)
    s2 = CS.summarize_sq("cfg2", 1, str(ps), src_hash="abc")
    (e,) = s2["kernels"].values()
    assert e["launches"] == 1 and e["valu_instr_per_butterfly"] == pytest.approx(12.0) and e["held_clock_GHz"] == pytest.approx(30000.0 / 14000)


def test_design_md_tables_are_what_the_profiles_say():
    """DESIGN.md section 4's two tables live between <!-- generated:... --> markers and are written by tools/design_table.py
    from profiles/<round>_*: regenerating them from the committed profile set must reproduce the committed text exactly
    (the document cannot drift from the files it cites)."""
    import importlib

    dt = importlib.import_module("design_table")
    s = open(os.path.join(ROOT, "DESIGN.md")).read()
    spread = s.split("ms per step** (", 1)[1].split(") |", 1)[0]
    again = dt.replace(dt.replace(s, "headline", dt.headline(spread)), "configs", dt.configs())
    assert again == s


def test_phase_stamp_reducer_on_synthetic_records():
    """tools/phase_stamps.py: [waves][128] records of s_memtime stamps -> per-phase cycle table.  Synthetic records with known
    phase lengths (LDS-DMA kernel, R = 3: stamps 0..9 per iteration; column kernel, R = 2: no stamp 1): steady-state iterations
    are averaged apart from iteration 0, shares sum to 1, the clock comes from the s_memrealtime pair (100 MHz)."""
    import numpy as np

    import phase_stamps as ps

    R, nst = 3, 10
    lens = [120, 600, 2800, 550, 2750, 430, 1650, 300, 100]  # stamp k-1 -> k, k = 1..9
    recs = np.zeros((6, 128), dtype=np.uint64)
    for w in range(4):  # two records stay empty (slot 1 == 0): not sampled
        r = recs[w]
        r[0], r[1], r[3] = 1000, 50000, 3
        t = 50000 + 3000  # init
        for it in range(3):
            scale = 2 if it == 0 else 1  # a cold first iteration
            r[4 + it * 12] = t
            for k in range(1, nst):
                t += lens[k - 1] * scale
                r[4 + it * 12 + k] = t
            t += 250  # loop back
        r[126] = t
        r[127] = 1000 + (t - 50000) // 20  # 100 MHz ticks at 2.0 GHz
    o = ps.reduce_region(recs, R, True, 8, 8, 256)
    assert o["waves_sampled"] == 4 and o["clock_GHz_median"] == pytest.approx(2.0, rel=1e-3) and o["init_cycles_median"] == 3000
    assert [p["cycles_mean"] for p in o["phases"]] == [pytest.approx(v) for v in lens + [250]]
    assert o["phases"][2]["cycles_mean_iteration0"] == pytest.approx(5600)
    assert o["iteration_cycles_mean_steady"] == pytest.approx(sum(lens) + 250)
    assert sum(p["share_of_iteration"] for p in o["phases"]) == pytest.approx(1.0)
    assert o["split"]["compute (register rounds)"]["cycles"] == pytest.approx(2800 + 2750 + 1650)
    assert o["split"]["exchange (LDS)"]["cycles"] == pytest.approx(550 + 430)
    assert o["wave_butterflies_per_iteration"] == 32 and o["wave_elapsed_cycles_per_wave_butterfly"] == pytest.approx((sum(lens) + 250) / 32)
    # a kernel without an LDS-DMA tile writes no stamp 1: its load phase runs from stamp 0 to stamp 2
    recs2 = np.zeros((2, 128), dtype=np.uint64)
    r = recs2[0]
    r[0], r[1], r[3] = 10, 1000, 2
    t = 2000
    for it in range(2):
        for k in range(8):
            if k != 1:
                r[4 + it * 12 + k] = t
            t += 100
        t += 40
    r[126], r[127] = t, 10 + (t - 1000) // 24
    o2 = ps.reduce_region(recs2, 2, False, 8, 16, 256)
    assert [p["ends_at_stamp"] for p in o2["phases"]] == [2, 3, 4, 5, 6, 7, 0]
    assert o2["phases"][0]["cycles_mean"] == pytest.approx(200) and o2["phases"][-1]["cycles_mean"] == pytest.approx(140)
    assert ps.reduce_region(np.zeros((3, 128), dtype=np.uint64), 2, False, 8, 16, 256)["waves_sampled"] == 0


def test_power_model_fit_reproduces_its_own_line_and_halves_cycle_savings():
    """tools/power_model.py: the three-term energy model is fitted to the committed power probe; it must reproduce the transform's
    own (clock, time), return roughly HALF of a cycle saving as time under the cap (MI355X_MICROARCH.md DVFS give-back; round 5's
    8-wave bound measured -2.9 % for -4.8 % cycles), and price a single HBM trip above any cycle saving on offer."""
    import glob

    import power_model as pm

    path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_power_probe.txt")))[-1]
    probe = pm.parse_probe(open(path).read())
    assert {"idle-ish (sync only)", "forward (real)", "forward (L2 loads, no stores)", "copy (xor kernel)"} <= set(probe)
    m = pm.fit(probe)
    f, t = pm.solve(m)
    assert f == pytest.approx(m["f_real_GHz"], rel=2e-3) and t == pytest.approx(m["t_real_ms"], rel=2e-3)
    assert 2.0 < m["alpha"] < 4.5 and m["E_mem_J"] > m["shares_at_operating_point"]["butterflies"]  # HBM trips: the largest share
    _, t10 = pm.solve(m, cycles_scale=0.9)
    assert 0.35 < (1 - t10 / t) / 0.10 < 0.65  # about half of a cycle saving returns as time
    _, t1trip = pm.solve(m, mem_scale=0.5)
    assert t1trip < pm.solve(m, cycles_scale=0.8)[1]
    # an un-capped case: no energy at all beyond idle -> the clock stays at f_max and time follows cycles
    free = dict(m, E_mem_J=0.0, E_valu_J_at_f0=0.0)
    f2, t2 = pm.solve(free, cycles_scale=0.5)
    assert f2 == 2.4 and t2 == pytest.approx(0.5 * m["cycles"] / 2.4e9 * 1e3)


def test_phase_stamp_chrome_trace_is_in_the_references_trace_format():
    """tools/phase_stamps.py --trace-prefix: one CU's waves as chrome-trace events with the keys of the reference's
    profile/trace/*.json (name, ts in cycles, ph B / E / M, pid, tid): B and E balance per row, time never runs backwards within
    a row, only the first recorded wave's CU is traced."""
    import numpy as np

    import phase_stamps as ps

    recs = np.zeros((6, 128), dtype=np.uint64)
    for w in range(4):
        r = recs[w]
        r[0], r[1], r[3] = 1000, 5000 + 10 * w, 2
        r[2] = (3 << 32) | (0x21 << 8) | ((w % 4) << 4) | (w // 4)  # XCC 3, CU byte 0x21, SIMD w, wave 0
        t = 6000 + 10 * w
        for it in range(2):
            for k in range(10):
                r[4 + it * 12 + k] = t
                t += 100
            t += 50
        r[126], r[127] = t, 1100
    recs[4, 1], recs[4, 2] = 7000, (5 << 32) | (0x33 << 8)  # a wave of ANOTHER CU: not traced
    ev = ps.chrome_trace(recs, 3, True)
    meta = [e for e in ev if e["ph"] == "M"]
    assert meta[0]["name"] == "process_name" and "XCC 3" in meta[0]["args"]["name"] and len(meta) == 1 + 4
    assert {e["args"]["name"] for e in meta[1:]} == {"SIMD %d wave 0" % k for k in range(4)}
    body = [e for e in ev if e["ph"] != "M"]
    assert all(set(e) == {"name", "ts", "ph", "pid", "tid", "args"} for e in body)
    for tid in range(4):
        row = [e for e in body if e["tid"] == tid]
        assert [e["ph"] for e in row] == ["B", "E"] * (len(row) // 2) and all(a["ts"] <= b["ts"] for a, b in zip(row, row[1:]))
        assert len(row) == 2 * (1 + 2 * 9)  # init + 9 phases x 2 iterations
        assert row[0]["name"].startswith("init") and row[2]["name"] == "wait" and any(e["name"].startswith("round 0") for e in row)
    assert ps.chrome_trace(np.zeros((2, 128), dtype=np.uint64), 3, True) == []
