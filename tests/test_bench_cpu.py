"""bench.py's host logic on the CPU: the per-rank verification reduce at world 2 and world 8 (one bad rank, or ranks that
share a device, turn the whole line red), the coefficient-sum invariant's arithmetic, the vector-ALU roofline on the SIMD-32
peak with a synthetic mix, the bound decision, the extra configurations' generators and checks, and the tools that feed them."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, spawn_world

sys.path.insert(0, os.path.join(ROOT, "tools"))

GOLD = 0xFFFFFFFF00000001


def test_rowsum_mod_p_is_exact_and_is_the_networks_output_zero(oracle):
    import bench

    rng = np.random.default_rng(4)
    rows = rng.integers(0, 2**63, size=(5, 1 << 10), dtype=np.uint64) % np.uint64(GOLD)
    rows[0, :] = GOLD - 1  # sums far beyond 64 bits
    want = [sum(int(v) for v in r) % GOLD for r in rows]
    assert bench.rowsum_mod_p(rows, GOLD) == want
    # int64 views (what a torch buffer hands over) carry the same bits
    assert bench.rowsum_mod_p(rows.view(np.int64), GOLD) == want
    # the invariant itself: out[0] of the reference network is the plain coefficient sum, whatever the table
    T = oracle.make_roots(1 << 10, GOLD, 7, 8)
    out = oracle.ntt(rows, T, GOLD)
    assert [int(v) for v in out[:, 0]] == want


def _reduce_worker(rank, world, rdzv, q, bad_rank, same_device):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import bench
    from conftest import init_gloo_or_report

    if not init_gloo_or_report(rank, world, rdzv, q):
        return
    ok = rank != bad_rank
    bus = 5 if same_device else 5 + rank  # same_device: every rank reports ONE GPU (a mis-bound launch)
    ident = {"rank": rank, "local_device": rank, "pci_bus_id": "0000:%02x:00.0" % bus, "ms_per_step": 1.5 + rank,
             "round_trip_identical": ok, "coefficient_sum_invariant": True}
    reduced, recs = bench.reduce_verdicts(dist, torch, torch.device("cpu"), world, rank, [ok, True], ident)
    fields, code = bench.verdict_fields(reduced, recs, dist.get_world_size())
    q.put((rank, code, fields))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,bad_rank", [(2, -1), (2, 1), (2, 0), (8, -1), (8, 5)])
def test_one_bad_rank_turns_the_line_red(world, bad_rank):
    """gloo, world 2 and world 8 (the node the driver scales to): every rank contributes (round trip ok, coefficient sum ok) + its
    identity; all_reduce(MIN) + all_gather_object of `world` identity records.  With every rank good: all_ranks_verified, exit code
    0.  With ONE bad rank -- whichever, one of eight included -- every rank learns it, rank 0's line says false and every process
    exits 1."""
    res = spawn_world(_reduce_worker, world, extra_args=(bad_rank, False))
    assert len(res) == world
    for rank, code, f in res:
        assert f["world_size_seen"] == world and [r["rank"] for r in f["ranks"]] == list(range(world))  # gathered in rank order
        assert f["distinct_devices"] == world and [r["ms_per_step"] for r in f["ranks"]] == [1.5 + r for r in range(world)]
        assert f["verification"]["ranks_on_distinct_devices"] is True
        if bad_rank < 0:
            assert code == 0 and f["all_ranks_verified"] is True
        else:
            assert code == 1 and f["all_ranks_verified"] is False
            assert f["verification"]["round_trip_identical_all"] is False and f["verification"]["coefficient_sum_invariant_all"] is True
            assert [r["round_trip_identical"] for r in f["ranks"]] == [r != bad_rank for r in range(world)]  # and WHICH rank it was


def test_ranks_on_one_device_turn_the_line_red():
    """A world-2 job whose ranks both report the same PCI id (every check passed!) is NOT verified: its aggregate would be one
    GPU's time-sliced work.  Only the one-device rehearsal switch waives the rule, and the line then says it was waived."""
    res = spawn_world(_reduce_worker, 2, extra_args=(-1, True))
    for rank, code, f in res:
        assert code == 1 and f["all_ranks_verified"] is False and f["distinct_devices"] == 1
        assert f["verification"]["round_trip_identical_all"] is True and f["verification"]["ranks_on_distinct_devices"] is False
    import bench

    recs = [{"rank": r, "pci_bus_id": "0000:05:00.0"} for r in range(2)]
    f, code = bench.verdict_fields([True, True], recs, 2, one_device_ok=True)
    assert code == 0 and f["all_ranks_verified"] and f["verification"]["distinctness_waived_one_device_rehearsal"] is True
    # a rank without any identity cannot be shown distinct: red as well
    f, code = bench.verdict_fields([True, True], [{"rank": 0, "pci_bus_id": "a"}, {"rank": 1, "pci_bus_id": None, "uuid": None}], 2)
    assert code == 1 and not f["all_ranks_verified"]
    # fewer records than ranks (a rank that never reported): red
    f, code = bench.verdict_fields([True, True], [{"rank": 0, "pci_bus_id": "a"}], 2)
    assert code == 1


def test_verdict_fields_single_rank():
    import bench

    rec = {"rank": 0, "local_device": 0, "pci_bus_id": None, "uuid": "GPU-abc", "ms_per_step": 1.6}
    f, code = bench.verdict_fields([True, True], [rec], 1)
    assert code == 0 and f["all_ranks_verified"] and f["distinct_devices"] == 1 and f["world_size_seen"] == 1
    f, code = bench.verdict_fields([True, False], [rec], 1)
    assert code == 1 and not f["all_ranks_verified"]
    f, code = bench.verdict_fields([True, True], [dict(rec, uuid=None)], 1)
    assert f["distinct_devices"] is None and code == 0  # no identity available: said, not guessed; one rank needs no distinctness


def test_valu_roofline_arithmetic_on_a_synthetic_mix():
    """roofline.valu on the unit's MEASURED throughput (tools/hw.py, profiles/r06_valu_peak.txt: 4 cycles per wave64 instruction for
    the VOP3-class forms of the butterfly statements, 2 for plain moves / adds): the forward statement's 22 VALU (20 + 2 moves) cost
    84 cycles at peak, the counter's extra instructions 4 each; at 103.0 / 104.9 kernel cycles per wave-butterfly per SIMD that is
    0.86 / 0.81 -- rounds 1-4's flat 4-cycle price said 0.91, round 5's first re-base on the guide's 2 cycles said 0.45."""
    import bench

    assert (bench.VALU_PEAK_CYCLES_VOP3, bench.VALU_PEAK_CYCLES_PLAIN, bench.SIMDS) == (4.0, 2.0, 1024)
    assert bench.valu_peak_cycles(22, 2) == 84.0 and bench.valu_peak_cycles(23.15, 2) == pytest.approx(88.6)
    passes = [("contig", 0, 8), ("col", 8, 8)]
    assert bench.stream_plain_counts(passes) == [(22, 2), (22, 2)]  # from the generator's own instruction lists
    bf_waves = 4096 * 32768 * 8 / 64  # wave-butterflies per launch
    kc = [103.0, 104.9]               # round 5's counters: kernel cycles per wave-butterfly per SIMD
    cyc = [k * bf_waves / 1024 for k in kc]
    ent = [("k0", {"valu_instr_per_butterfly": 23.15, "held_clock_GHz": 1.98, "kernel_cycles": cyc[0], "mean_waves_per_simd": 3.7,
                   "wave_issue_stall_frac": 0.47}),
           ("k1", {"valu_instr_per_butterfly": 22.20, "held_clock_GHz": 2.02, "kernel_cycles": cyc[1], "mean_waves_per_simd": 3.6,
                   "wave_issue_stall_frac": 0.41})]
    st = {"cycles_per_butterfly_by_waves_per_simd": {1: 143.8, 2: 97.2, 3: 96.6, 4: 82.6, 8: 81.0}, "source": "synthetic"}
    v = bench.valu_roofline(ent, passes, [0.83, 0.80], 4096, 16, stream_counts=[(22, 2), (22, 2)], statement=st)
    assert v["peak_cycles_per_wave_instr"]["vop3_class"] == 4.0 and v["peak_cycles_per_wave_instr"]["plain_moves_adds"] == 2.0
    assert v["peak_cycles_per_butterfly"] == [pytest.approx(88.6), pytest.approx(84.8)]
    assert v["frac_of_peak_at_held_clock_per_pass"] == [pytest.approx(88.6 / 103.0), pytest.approx(84.8 / 104.9)]  # 0.86 / 0.81
    assert v["frac_of_peak_at_held_clock"] == pytest.approx((88.6 + 84.8) / (103.0 + 104.9))
    assert v["kernel_cycles_per_wave_butterfly_per_simd"] == [pytest.approx(103.0), pytest.approx(104.9)]
    assert v["peak_butterflies_per_s"] == pytest.approx(1024 * 2.4e9 / 86.7 * 64)
    assert v["saturated"] is False and "verdict" not in v and "what" not in v  # (round 6: numbers only; the words are DESIGN.md section 4)
    sa = v["statement_alone_steady_state"]
    assert sa["cycles_per_butterfly_at_4_or_more_waves"] == 81.0 and sa["kernel_over_statement"] == [pytest.approx(103.0 / 81.0), pytest.approx(104.9 / 81.0)]
    for old in ("frac_at_held_clock", "frac_at_held_clock_weighted", "frac_at_2.4GHz", "issue_cost_at_kernel_occupancy"):
        assert old not in v  # earlier rounds' keys are gone, not aliased
    # without the statement's mix EVERY counted instruction is priced at 4 cycles: an upper estimate
    u = bench.valu_roofline(ent, passes, [0.83, 0.80], 4096, 16)
    assert u["peak_cycles_per_butterfly"] == [pytest.approx(23.15 * 4), pytest.approx(22.2 * 4)] and "statement_alone_steady_state" not in u
    assert u["frac_of_peak_at_held_clock"] > v["frac_of_peak_at_held_clock"]
    # a kernel that really sat at the unit's throughput would say so
    fast = [(k, dict(e, kernel_cycles=bench.valu_peak_cycles(e["valu_instr_per_butterfly"], 2) / 0.97 * bf_waves / 1024)) for k, e in ent]
    assert bench.valu_roofline(fast, passes, [0.4, 0.4], 4096, 16, stream_counts=[(22, 2), (22, 2)])["saturated"] is True


def test_statement_steady_state_is_parsed_from_the_probe_file(tmp_path, monkeypatch):
    import bench

    prof = tmp_path / "profiles"
    prof.mkdir()
    (tmp_path / "tools").mkdir()
    (tmp_path / "tools" / "gen_gl_asm.py").write_text("# generator\n")
    (tmp_path / "tools" / "stream_occupancy.hip").write_text("// probe\n")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    body = ("# closed batch\n1  141.68  6.440  2.304  141.69\n4  72.21  3.282  2.395  92.43\n"
            "# steady state: 12 generations of workgroups\n# waves/SIMD  cycles\n1  143.79  6.536  2.390\n4  82.64  3.756  2.372\n8  81.02  3.683  2.376\n")
    # round 6: quoted only when the file is stamped with the hash of THIS tree's generator + probe (tools/run_round.sh writes the line)
    (prof / "r05_stream_occupancy.txt").write_text(body)
    assert bench.statement_steady_state() is None  # unstamped: not re-measured on this tree
    (prof / "r06_stream_occupancy.txt").write_text("# stream_src_hash 0123456789abcdef\n" + body)
    assert bench.statement_steady_state() is None  # stamped with another tree's hash
    (prof / "r04_stream_occupancy.txt").write_text("# stream_src_hash %s\n" % bench.stream_source_hash() + body)
    st = bench.statement_steady_state()
    assert st["cycles_per_butterfly_by_waves_per_simd"] == {1: 143.79, 4: 82.64, 8: 81.02} and "r04_stream_occupancy.txt" in st["source"]
    (tmp_path / "tools" / "gen_gl_asm.py").write_text("# generator, changed\n")
    assert bench.statement_steady_state() is None


def test_config_plain_share_comes_from_the_generator():
    """The extra configurations' VALU fraction prices plain moves / adds at 2 cycles and the rest at 4: the share comes from the
    generator's instruction lists (Goldilocks 2 of 22; a 32-bit prime's carry statement 1 of 11; a lazy prime's 5 of 9)."""
    import bench

    assert bench.config_plain_share({"wb": 8, "p": GOLD}) == pytest.approx(2 / 22)
    assert bench.config_plain_share({"wb": 4, "p": 3221225473}) == pytest.approx(1 / 11)
    assert bench.config_plain_share({"wb": 4, "p": 998244353}) == pytest.approx(5 / 9)
    assert 0 < bench.config_plain_share({"wb": 8, "p": 0x3FFFFFEE00000001}) < 0.2


def test_decide_bound_from_the_runs_numbers():
    """roofline.bound is computed, not asserted: "hbm" / "valu" only at >= 0.9 of that roofline; "power-cap" when neither is
    saturated and the kernels hold < 0.9 of the peak clock; "unsaturated" when nothing can be shown."""
    import bench

    assert bench.SATURATED == 0.95
    assert bench.decide_bound([0.97, 0.96], 0.46)[0] == "hbm"
    assert bench.decide_bound([0.88, 0.89], 0.96)[0] == "valu"
    assert bench.decide_bound([0.93, 0.91], 0.46, [3.7, 3.7], [1.98, 2.02])[0] == "power-cap"  # 0.9-0.95 of a copy is not "hbm": the label must not flip with the copy's own noise
    b, why = bench.decide_bound([0.88, 0.89], 0.84, [3.7, 3.7], [1.95, 1.97])
    assert b == "power-cap" and "0.88" in why and "0.84" in why and "3.7 waves" in why and "1.95" in why
    b, why = bench.decide_bound([0.88, 0.89], 0.46, [3.7, 3.7], [2.38, 2.39])  # the clock is held: no cap to blame
    assert b == "unsaturated"
    b, why = bench.decide_bound([0.88, 0.89], None)
    assert b == "unsaturated" and "n/a" in why  # no counters for these sources: nothing is claimed
    assert bench.decide_bound(None, 0.96)[0] == "valu"


def test_synthetic_generators_and_config_checks(oracle):
    """The extra configurations' host logic: the 32-bit generator equals splitmix64 mod p on Python integers; the coefficient-sum
    invariant for 4-byte words; evaluation at a root of x^N + 1 accepts the oracle's negacyclic product and rejects a wrong word."""
    import torch

    import bench

    p, n = 3221225473, 64
    got = bench.synth_u32(torch, 3, n, p, torch.device("cpu"), first_row=7).numpy().view(np.uint32)

    def sm(i):
        z = (i + 0x9E3779B97F4A7C15) & (2**64 - 1)
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
        return z ^ (z >> 31)

    want = [[sm(bench.SEED + (7 + b) * n + i) % p for i in range(n)] for b in range(3)]
    assert got.tolist() == want
    assert bench.rowsum_mod_p_u32(got, p) == [sum(r) % p for r in want]
    T = oracle.make_roots(n, p, 5, 4)
    assert [int(v) for v in oracle.ntt(got, T, p)[:, 0]] == [sum(r) % p for r in want]
    # negacyclic product by schoolbook, checked by evaluation at r = g^((p-1)/2N), r^N = -1
    P, N = GOLD, 32
    rng = np.random.default_rng(9)
    a = (rng.integers(0, 2**63, size=N, dtype=np.uint64) % np.uint64(P))
    b = (rng.integers(0, 2**63, size=N, dtype=np.uint64) % np.uint64(P))
    c = [0] * N
    for i in range(N):
        for j in range(N):
            t = int(a[i]) * int(b[j])
            if i + j < N:
                c[i + j] = (c[i + j] + t) % P
            else:
                c[i + j - N] = (c[i + j - N] - t) % P
    r = pow(7, (P - 1) // (2 * N), P)
    assert pow(r, N, P) == P - 1
    cv = np.array(c, dtype=np.uint64)
    assert bench.poly_eval_mod(cv, r, P) == bench.poly_eval_mod(a, r, P) * bench.poly_eval_mod(b, r, P) % P
    cv[5] ^= np.uint64(1)
    assert bench.poly_eval_mod(cv, r, P) != bench.poly_eval_mod(a, r, P) * bench.poly_eval_mod(b, r, P) % P


def test_configs_verdict_wrong_result_is_red_failed_leg_is_reported():
    """BASELINE configs 2 and 4 in the driver-run line: a WRONG result turns the exit code red; a leg that could not run is
    reported (not verified, with its error) without taking the headline down."""
    import bench

    ok = {"key": "cfg2", "verified": True}
    assert bench.configs_verdict([ok, dict(ok, key="cfg4")]) == (True, 0)
    assert bench.configs_verdict([ok, {"key": "cfg4", "verified": False, "verification": {"evaluation_at_root_of_xN_plus_1": False}}]) == (False, 1)
    assert bench.configs_verdict([ok, {"key": "cfg4", "verified": False, "error": "OutOfMemoryError()"}]) == (False, 0)
    assert bench.configs_verdict([]) == (True, 0)


def test_tagged_profile_picks_the_newest_collection_on_these_sources(tmp_path, monkeypatch):
    """Counters are quoted from the newest profiles/rNN_<name>.json whose stamped kernel-source hash equals the tree's -- an older
    round's file when this round did not touch the kernels, nothing at all when no file matches."""
    import bench

    prof = tmp_path / "profiles"
    prof.mkdir()
    for rnd, h in (("r03", "aaa"), ("r04", "bbb"), ("r05", "ccc")):
        (prof / ("%s_pmc_traffic.json" % rnd)).write_text(json.dumps({"src_hash": h, "round": rnd}))
    (prof / "r04_cfg2_pmc_traffic.json").write_text(json.dumps({"src_hash": "bbb", "round": "cfg2"}))  # another name: never confused
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    d, why = bench.tagged_profile("pmc_traffic", "bbb")
    assert d["round"] == "r04" and "r04_pmc_traffic.json" in why
    d, why = bench.tagged_profile("pmc_traffic", "ccc")
    assert d["round"] == "r05"
    d, why = bench.tagged_profile("pmc_traffic", "zzz")
    assert d is None and "not quoted" in why and "r05_pmc_traffic.json: ccc" in why
    d, why = bench.tagged_profile("sq_counters", "bbb")
    assert d is None and "absent" in why


def test_stream_mix_comes_from_the_generator():
    """tools/valu_mix.py: the Goldilocks forward butterfly is 22 VALU (4 multiply-adds, 14 carry-chain steps, one 64-bit
    compare, one select, two moves) + 5 scalar mask ops; the inverse 21 + 4 (no compare); the column pass reads its twiddles
    from SGPRs.  class_costs() averages the measured forms of a class at the asked occupancy and ignores mixed probes."""
    import valu_mix

    f = valu_mix.stream_mix("fwd")
    assert f == {"valu": 22, "salu": 5, "mix": {"carry": 14, "cmp64": 1, "cndmask": 1, "mad64": 4, "plain": 2}}
    assert valu_mix.stream_mix("fwd", True)["mix"]["mad64_s"] == 4
    i = valu_mix.stream_mix("inv")
    assert i["valu"] == 21 and i["salu"] == 4 and "cmp64" not in i["mix"]
    assert valu_mix.stream_mix("mul")["valu"] == 13
    ic = {"forms": {"a": {"class": "carry", "cycles": {"4": 4.0, "8": 4.2}}, "b": {"class": "carry", "cycles": {"4": 4.4, "8": 4.2}},
                    "c": {"class": "mad64", "cycles": {"4": 5.3, "8": 5.3}}, "d": {"class": "carry+salu", "cycles": {"4": 9.0, "8": 9.0}}}}
    assert valu_mix.class_costs(ic, 4) == {"carry": pytest.approx(4.2), "mad64": pytest.approx(5.3)}
    assert valu_mix.weighted_cycles({"carry": 2, "mad64": 1, "zzz": 1}, valu_mix.class_costs(ic, 4)) == pytest.approx(8.4 + 5.3 + 4.0)


# ---- round 6: the line is numbers, not commentary; impossible clocks are nulled; a hung rank cannot hang the job ------------
def test_sane_clocks_and_time_weighted_bound():
    """A GRBM_GUI_ACTIVE / 8 / duration quotient above the part's 2.4 GHz is not a clock (round 5's line carried 3.30 GHz for a
    14 us launch): nulled with a reason.  decide_bound weighs the kernels' clocks by their time: one short kernel a few per cent
    under the threshold no longer decides `power-cap`, and the detail says the clock is the counter run's."""
    import bench

    clocks, note = bench.sane_clocks([3.2969])
    assert clocks == [None] and "too short" in note
    clocks, note = bench.sane_clocks([1.93, 1.97, None])
    assert clocks == [1.93, 1.97, None] and note is None
    assert bench.sane_clocks([2.41])[0] == [2.41]  # within the tolerance of the quotient itself
    # config 4's shape in round 5: 2.07 / 2.14 / 2.24 GHz with the long middle kernel at 2.24 -- mean clock 2.19 >= 2.16: not a cap
    b, why = bench.decide_bound([0.54], 0.83, [3.6, 3.9, 3.6], [2.07, 2.14, 2.24], [2.5, 2.4, 7.1])
    assert b == "unsaturated" and "0.54" in why and "0.83" in why
    # ... while the plain minimum would have said power-cap; equal weights still do when every kernel is low
    assert bench.decide_bound([0.88, 0.89], 0.84, [3.7, 3.7], [1.95, 1.97], [0.85, 0.83])[0] == "power-cap"
    b, why = bench.decide_bound([0.88, 0.89], 0.84, [3.7, 3.7], [1.95, 1.97])
    assert b == "power-cap" and "counter run" in why and "1.95" in why


def test_slim_line_drops_prose_keeps_numbers():
    """bench.slim_line: the default line has no explanatory keys, provenance strings are cut to the file they name, nested floats
    carry 5 significant digits, the extra configurations keep their measurements (their shapes are tools/configs.py's, by key), and
    the contract's top-level numbers are untouched."""
    import json

    import bench

    full = {"metric": "m", "value": 2448329.4740061713, "ms_per_step": 1.6729774499253836, "unit": "NTT/s",
            "config": {"workload": "w", "data_note": "a[b][i] = ..."},
            "verification": {"round_trip_identical_all": True, "what": "x" * 300},
            "roofline": {"bound": "power-cap", "bound_detail": "short", "bound_note": "y" * 500, "definition": "z" * 400,
                         "frac": 0.32367429514351825, "pass_ms": [0.8301234567, 0.8171234567],
                         "traffic_source": "profiles/r05_pmc_traffic.json (src_hash abc); forward kernels: " + "k" * 200,
                         "valu": {"what": "v" * 600, "held_clock_GHz": [1.93, None],
                                  "kernels": ["void ntt::(anonymous namespace)::pass_kernel<ntt::PassCfg<ntt::FieldGL, 8, 0, true, false, 15, 3, 8, true>, false>(ntt::PassArgs<...>)"]}},
            "configs": [{"name": "c", "key": "cfg2", "ms": 0.014, "roofline": {"definition": "d" * 100, "bound_detail": "b" * 100, "frac": 0.29912345678}}]}
    slim = bench.slim_line(full)
    text = json.dumps(slim)
    assert slim["value"] == full["value"] and slim["ms_per_step"] == full["ms_per_step"]
    assert "what" not in slim["verification"] and "bound_note" not in slim["roofline"] and "definition" not in slim["roofline"]
    assert "definition" not in slim["configs"][0]["roofline"] and "what" not in slim["roofline"]["valu"] and "data_note" not in slim["config"]
    assert slim["roofline"]["frac"] == 0.32367 and slim["roofline"]["pass_ms"] == [0.83012, 0.81712] and slim["configs"][0]["roofline"]["frac"] == 0.29912
    assert slim["roofline"]["traffic_source"] == "r05_pmc_traffic.json" and slim["roofline"]["bound_detail"] == "short"  # (a file under profiles/)
    assert "kernels" not in slim["roofline"]["valu"] and slim["roofline"]["valu"]["held_clock_GHz"] == [1.93, None]
    assert "name" not in slim["configs"][0] and "bound_detail" not in slim["configs"][0]["roofline"]
    assert len(text) < 900 and full["roofline"]["bound_note"] == "y" * 500  # the input is not modified


def test_self_launch_ends_a_hung_job(tmp_path, monkeypatch):
    """`python bench.py --gpus N` starts its own ranks; a rank that hangs (instead of exiting) used to hang the parent for ever.
    With the wall-clock limit the parent terminates exactly the children it started and exits 124."""
    import argparse
    import time

    import bench

    hang = tmp_path / "hang.py"
    hang.write_text("import time, sys\nopen(sys.argv[-1] + '.started', 'a').close()\ntime.sleep(120)\n")
    monkeypatch.setattr(bench, "__file__", str(hang))
    monkeypatch.setenv("NTT_BENCH_ONE_DEVICE", "1")  # rehearsal: no device count is asked for
    monkeypatch.setenv("NTT_BENCH_LAUNCH_TIMEOUT_S", "2")
    t0 = time.monotonic()
    rc = bench.self_launch(argparse.Namespace(gpus=2), [])
    assert rc == 124 and time.monotonic() - t0 < 40
