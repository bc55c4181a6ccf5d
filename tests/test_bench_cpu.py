"""bench.py's host logic on the CPU: the per-rank verification reduce (one bad rank turns the whole line red), the
coefficient-sum invariant's arithmetic, the weighted VALU issue model on a synthetic mix, and the tools that feed it."""
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


def _reduce_worker(rank, world, port, q, bad_rank):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import bench
    from conftest import init_gloo_or_report

    if not init_gloo_or_report(rank, world, port, q):
        return
    ok = rank != bad_rank
    ident = {"rank": rank, "local_device": rank, "pci_bus_id": "0000:%02x:00.0" % (5 + rank), "ms_per_step": 1.5 + rank,
             "round_trip_identical": ok, "coefficient_sum_invariant": True}
    reduced, recs = bench.reduce_verdicts(dist, torch, torch.device("cpu"), world, rank, [ok, True], ident)
    fields, code = bench.verdict_fields(reduced, recs, dist.get_world_size())
    q.put((rank, code, fields))
    dist.destroy_process_group()


@pytest.mark.parametrize("bad_rank", [-1, 1, 0])
def test_one_bad_rank_turns_the_line_red(bad_rank):
    """world-2 gloo: every rank contributes (round trip ok, coefficient sum ok) + its identity; all_reduce(MIN) + all_gather.
    With every rank good: all_ranks_verified, exit code 0.  With ONE bad rank -- whichever -- every rank learns it, rank 0's
    line says false and every process exits 1."""
    res = spawn_world(_reduce_worker, 2, extra_args=(bad_rank,))
    for rank, code, f in res:
        assert f["world_size_seen"] == 2 and [r["rank"] for r in f["ranks"]] == [0, 1]  # gathered in rank order
        assert f["distinct_devices"] == 2 and [r["ms_per_step"] for r in f["ranks"]] == [1.5, 2.5]
        if bad_rank < 0:
            assert code == 0 and f["all_ranks_verified"] is True
        else:
            assert code == 1 and f["all_ranks_verified"] is False
            assert f["verification"]["round_trip_identical_all"] is False and f["verification"]["coefficient_sum_invariant_all"] is True
            assert [r["round_trip_identical"] for r in f["ranks"]] == [bad_rank != 0, bad_rank != 1]  # and WHICH rank it was


def test_verdict_fields_single_rank():
    import bench

    rec = {"rank": 0, "local_device": 0, "pci_bus_id": None, "uuid": "GPU-abc", "ms_per_step": 1.6}
    f, code = bench.verdict_fields([True, True], [rec], 1)
    assert code == 0 and f["all_ranks_verified"] and f["distinct_devices"] == 1 and f["world_size_seen"] == 1
    f, code = bench.verdict_fields([True, False], [rec], 1)
    assert code == 1 and not f["all_ranks_verified"]
    f, _ = bench.verdict_fields([True, True], [dict(rec, uuid=None)], 1)
    assert f["distinct_devices"] is None  # no identity available: said, not guessed


def test_weighted_issue_model_arithmetic():
    """roofline.valu with measured per-class issue costs on a synthetic mix: 22-instruction stream of
    4 v_mad_u64_u32 at 5 cycles, 16 carry / compare / select forms at 4 and 2 plain moves at 2 = 88 cycles -- the same as a
    flat 4 -- plus 1.5 overhead instructions at 4: the weighted fraction equals the flat one there, and moves with the costs."""
    import bench

    mix = {"valu": 22, "salu": 5, "mix": {"mad64": 4, "carry": 14, "cmp64": 1, "cndmask": 1, "plain": 2}}
    costs = {"mad64": 5.0, "carry": 4.0, "cmp64": 4.0, "cndmask": 4.0, "plain": 2.0, "other": 4.0}
    assert bench.weighted_issue_cycles(mix, 22.0, costs) == pytest.approx(88.0)
    assert bench.weighted_issue_cycles(mix, 23.5, costs) == pytest.approx(88.0 + 1.5 * 4.0)
    assert bench.weighted_issue_cycles(mix, 21.0, costs) == pytest.approx(88.0)  # never negative overhead
    assert bench.weighted_issue_cycles({"valu": 1, "mix": {"unknown": 1}}, 1.0, costs) == pytest.approx(4.0)  # unpriced class: 4
    passes = [("contig", 0, 8), ("col", 8, 8)]
    bf_waves = 4096 * 32768 * 8 / 64
    cyc = [1.6e6, 1.5e6]
    ent = [("k0", {"valu_instr_per_butterfly": 23.5, "held_clock_GHz": 1.93, "kernel_cycles": cyc[0],
                   "valu_instr_x4cyc_over_kernel_cycles": 23.5 * 4 * bf_waves / (1024 * cyc[0]), "wave_issue_stall_frac": 0.46}),
           ("k1", {"valu_instr_per_butterfly": 22.0, "held_clock_GHz": 1.96, "kernel_cycles": cyc[1],
                   "valu_instr_x4cyc_over_kernel_cycles": 22.0 * 4 * bf_waves / (1024 * cyc[1]), "wave_issue_stall_frac": 0.41})]
    model = {"costs": costs, "streams": [mix, mix], "overhead_cycles": 4.0, "source": "synthetic"}
    v = bench.valu_roofline(ent, passes, [0.83, 0.80], 4096, 16, issue_model=model)
    assert v["issue_cycles_per_butterfly_weighted"] == [pytest.approx(94.0), pytest.approx(88.0)]
    assert v["frac_at_held_clock_weighted"] == pytest.approx(v["frac_at_held_clock"])  # 88 = 22 x 4: same price in total
    assert v["frac_at_held_clock_weighted_per_pass"][0] == pytest.approx(94.0 * bf_waves / (1024 * cyc[0]))
    # dearer multiplies: the weighted figure rises, the flat one cannot
    dear = dict(costs, mad64=6.0)
    w = bench.valu_roofline(ent, passes, [0.83, 0.80], 4096, 16, issue_model=dict(model, costs=dear))
    assert w["frac_at_held_clock_weighted"] > v["frac_at_held_clock_weighted"] and w["frac_at_held_clock"] == v["frac_at_held_clock"]
    assert isinstance(w["saturated"], bool) and ("saturated" in w["verdict"] or "stalls" in w["verdict"])
    # without a model nothing weighted is claimed
    assert "frac_at_held_clock_weighted" not in bench.valu_roofline(ent, passes, [0.83, 0.80], 4096, 16)


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
