#!/usr/bin/env python3
"""Where the cycles of the two headline pass kernels go: s_memtime stamps at every phase boundary of run_pass() (pass.h:
stamp()), written by lane 0 of EVERY wave of a launch of the diagnostic build ab/libntt_stamps.so
(`tools/ab_build.sh stamps -DNTT_PHASE_STAMPS`; the product and experiment libraries contain no stamp), reduced to a per-phase
cycle table.  The reference's analogue: the per-event hardware trace of one tile bracketed by event0()/event1()
(src/aie_core.cc:129-131, profile/trace/trace_16core_n11.json -- 63 % LockStall).

usage (GPU box): python3 tools/phase_stamps.py [--logn 16] [--batch 4096] [--reps 10] > profiles/rNN_phase_stamps.json
                 python3 tools/phase_stamps.py --logn 12 --p 3221225473 --g 5 --word-bytes 4 --batch 1024 --shape 4,0,8,512 > profiles/rNN_phase_stamps_cfg2.json

What is measured: the forward transform at the headline shape, `reps` back-to-back launches with stamps on (the last launch's
records are read).  A stamp waits for the wave's outstanding LDS traffic (and, in kernels that load straight into registers,
its global loads) before it reads the clock, so a phase ends when its data has arrived; stamping itself costs cycles -- the
`overhead` entry compares the stamped launch's duration with the product library's in the same process.  Outputs of the stamped
launch are compared word for word with the product library's."""
import argparse
import ctypes as C
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402

REC, HDR, PER_IT = 128, 4, 12  # pass.h: STAMP_RECORD, STAMP_HEADER, STAMPS_PER_ITER


def phase_names(R, dma):
    """name of the interval that ENDS at stamp k (k = 1 .. 2R+3)"""
    names = {1: "wait: tile landed in LDS (LDS-DMA prefetch issued one iteration earlier)" if dma else "(no separate tile stamp)",
             2: ("issue the next tile's LDS-DMA prefetch + read round 0's words from LDS" if dma else
                 "global loads straight into the round registers, issued and waited for")}
    for r in range(R):
        names[3 + 2 * r] = "round %d butterflies (register stages)" % r
        if r < R - 1:
            names[4 + 2 * r] = "exchange %d -> %d through LDS (write, sync, read)" % (r, r + 1)
    names[2 * R + 2] = "canonicalise + issue the stores"
    names[2 * R + 3] = "end-of-iteration sync"
    return names


def group_of(name):
    if name.startswith("round"):
        return "compute (register rounds)"
    if name.startswith("exchange"):
        return "exchange (LDS)"
    if "sync" in name or "loop" in name:
        return "sync + loop"
    return "memory (tile wait / loads / stores)"


def reduce_region(recs, R, dma, log_m, e_words, nt):
    """recs: [records][128] uint64 of one pass kind -> the per-phase table: steady-state iterations (it >= 1), iteration 0 apart.
    Pure numpy (CPU unit test: tests/test_profile_tools.py)."""
    recs = np.asarray(recs, dtype=np.uint64)
    live = recs[recs[:, 1] != 0]
    out = {"waves_sampled": int(len(live)), "rounds": R, "stages": log_m, "threads_per_workgroup": nt, "words_per_thread": e_words}
    if not len(live):
        return out
    iters = live[:, 3].astype(np.int64)
    life = (live[:, REC - 2] - live[:, 1]).astype(np.float64)
    real = (live[:, REC - 1] - live[:, 0]).astype(np.float64)  # s_memrealtime: 100 MHz
    out["clock_GHz_median"] = float(np.median(life / np.maximum(1.0, real) * 0.1))
    out["iterations_per_wave"] = {"min": int(iters.min()), "median": float(np.median(iters)), "max": int(iters.max())}
    if int(iters.max()) > 8:  # pass.h STAMP_ITERS: a record holds 8 iterations, the device stamps later ones into a scratch slot
        out["iterations_recorded"] = "the first 8 of each wave; later iterations ran and were stamped into the record's scratch slot"
    out["wave_lifetime_cycles_median"] = float(np.median(life))
    # the launch on the 100 MHz wall clock (s_memrealtime, one counter for the whole chip): when the waves START (the dispatcher's
    # ramp) against how long one of them lives -- a one-generation launch is the sum of the two, not a throughput
    start, end = live[:, 0].astype(np.float64), live[:, REC - 1].astype(np.float64)
    out["wall_clock_us"] = {"first_wave_start_to_last_wave_end": float(end.max() - start.min()) / 100.0,
                            "wave_starts_span_p01_p99": float(np.percentile(start, 99) - np.percentile(start, 1)) / 100.0,
                            "wave_lifetime_median": float(np.median(end - start)) / 100.0}
    # kernel entry -> the first iteration's first stamp: index set-up + the resident twiddle loads
    out["init_cycles_median"] = float(np.median(live[:, HDR].astype(np.float64) - live[:, 1].astype(np.float64)))
    nst = 2 * R + 4
    names = phase_names(R, dma)
    steady = {k: [] for k in range(1, nst)}
    cold = {k: [] for k in range(1, nst)}
    gaps, it_total, it0_total = [], [], []
    for row, n_it in zip(live, iters):
        t = row[HDR:HDR + 8 * PER_IT].astype(np.float64).reshape(8, PER_IT)
        if not dma:
            t[:, 1] = t[:, 0]  # kernels without an LDS-DMA tile write no stamp 1: the load phase runs from stamp 0 to stamp 2
        for it in range(min(int(n_it), 8)):
            d = np.diff(t[it, :nst])
            tgt = cold if it == 0 else steady
            for k in range(1, nst):
                tgt[k].append(d[k - 1])
            (it0_total if it == 0 else it_total).append(t[it, nst - 1] - t[it, 0])
            if it + 1 < n_it and it + 1 < 8:
                gaps.append(t[it + 1, 0] - t[it, nst - 1])
    if not it_total:  # one iteration per wave: report that one
        steady, it_total = cold, it0_total
        out["note"] = "every wave ran ONE iteration: the table is iteration 0"
    bf_per_thread = (e_words // 2) * log_m  # butterflies per thread per iteration = wave-butterflies per wave-iteration
    gap = float(np.mean(gaps)) if gaps else 0.0
    tot = float(np.mean(it_total)) + gap
    phases = []
    for k in range(1, nst):
        if not dma and k == 1:
            continue
        m = float(np.mean(steady[k]))
        phases.append({"ends_at_stamp": k, "phase": names[k], "cycles_mean": m, "cycles_median": float(np.median(steady[k])),
                       "share_of_iteration": m / tot, "cycles_mean_iteration0": float(np.mean(cold[k])) if cold[k] else None})
    if gaps:
        phases.append({"ends_at_stamp": 0, "phase": "loop back to the next iteration's first stamp", "cycles_mean": gap,
                       "cycles_median": float(np.median(gaps)), "share_of_iteration": gap / tot, "cycles_mean_iteration0": None})
    out["iteration_cycles_mean_steady"] = tot
    out["iteration_cycles_mean_first"] = float(np.mean(it0_total)) if it0_total else None
    out["wave_butterflies_per_iteration"] = bf_per_thread
    out["wave_elapsed_cycles_per_wave_butterfly"] = tot / bf_per_thread
    grp = {}
    for ph in phases:
        grp[group_of(ph["phase"])] = grp.get(group_of(ph["phase"]), 0.0) + ph["cycles_mean"]
    out["split"] = {k: {"cycles": v, "share": v / tot} for k, v in grp.items()}
    out["phases"] = phases
    return out


def chrome_trace(recs, R, dma, window_cycles=300000, label="pass"):
    """Chrome-trace events (the format of the reference's profile/trace/*.json: name / ts / ph B|E / pid / tid, ts in CYCLES) for the
    waves of ONE compute unit -- the analogue of the reference's one traced tile (src/aie2.py:157-158, tile (0,0)).  The CU is the one
    the first recorded wave ran on (HW_ID: cu, sh, se + XCC id); rows are its wave slots (SIMD, wave id), events the phases between
    two stamps, the first `window_cycles` cycles of the launch on that CU.  Pure numpy (CPU unit test)."""
    recs = np.asarray(recs, dtype=np.uint64)
    live = recs[recs[:, 1] != 0]
    if not len(live):
        return []
    hw = live[:, 2]
    cu_key = ((hw >> np.uint64(8)) & np.uint64(0xFF)) | ((hw >> np.uint64(32)) << np.uint64(8))  # CU_ID | SH_ID | SE_ID, XCC above
    mine = live[cu_key == cu_key[0]]
    t0 = float(mine[:, 1].min())
    names = phase_names(R, dma)
    nst = 2 * R + 4
    slots = sorted(set(int(h & np.uint64(0x3F)) for h in mine[:, 2]))  # SIMD_ID [5:4] | WAVE_ID [3:0]
    ev = [{"name": "process_name", "ph": "M", "pid": 0, "args": {"name": "%s: phase stamps of one CU (XCC %d, HW_ID[15:8] 0x%02x)" % (
        label, int(mine[0, 2] >> np.uint64(32)) & 15, int(mine[0, 2] >> np.uint64(8)) & 0xFF)}}]
    for i, sl in enumerate(slots):
        ev.append({"name": "thread_name", "ph": "M", "pid": 0, "tid": i, "args": {"name": "SIMD %d wave %d" % (sl >> 4, sl & 15)}})
    tid_of = {sl: i for i, sl in enumerate(slots)}
    for row in mine[np.argsort(mine[:, 1])]:
        tid = tid_of[int(row[2] & np.uint64(0x3F))]
        start = float(row[1]) - t0
        if start > window_cycles:
            continue
        t = row[HDR:HDR + 8 * PER_IT].astype(np.float64).reshape(8, PER_IT) - t0
        ev.append({"name": "init (indices, resident twiddles)", "ts": start, "ph": "B", "pid": 0, "tid": tid, "args": {}})
        ev.append({"name": "init (indices, resident twiddles)", "ts": float(t[0, 0]), "ph": "E", "pid": 0, "tid": tid, "args": {}})
        for it in range(min(int(row[3]), 8)):
            prev = 0
            for k in range(1, nst):
                if not dma and k == 1:
                    continue
                nm = names[k].split(" (")[0].split(":")[0]
                ev.append({"name": nm, "ts": float(t[it, prev]), "ph": "B", "pid": 0, "tid": tid, "args": {"iteration": it}})
                ev.append({"name": nm, "ts": float(t[it, k]), "ph": "E", "pid": 0, "tid": tid, "args": {}})
                prev = k
    return ev


def main():
    import torch

    from bench import GOLDILOCKS, synth_batch, synth_u32
    from ntt_aie_amd import _lib

    ap = argparse.ArgumentParser()
    ap.add_argument("--logn", type=int, default=16)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--p", type=int, default=GOLDILOCKS)
    ap.add_argument("--g", type=int, default=7)
    ap.add_argument("--word-bytes", type=int, default=8)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--shape", action="append", default=None,
                    help="per pass, in order: R,dma,E,NT (register rounds, 1 if the kernel has an LDS-DMA tile, words per thread, threads per "
                         "workgroup).  Default = the headline's two kernels: 3,1,8,256 and 2,0,16,256.  BASELINE config 2 (--logn 12 --p "
                         "3221225473 --g 5 --word-bytes 4 --batch 1024): --shape 4,0,8,512")
    ap.add_argument("--lib", default=os.path.join(ROOT, "ab", "libntt_stamps.so"))
    ap.add_argument("--trace-prefix", default=None,
                    help="also write <prefix>_<pass>.json: chrome-trace events (the reference's profile/trace format) of ONE compute unit's waves, "
                         "the first --trace-cycles cycles of the launch")
    ap.add_argument("--trace-cycles", type=int, default=300000)
    ap.add_argument("--dbg", type=int, default=0,
                    help="an experiment + stamps build only (make exp EXTRA=-DNTT_PHASE_STAMPS): ntt_plan_set_debug(FLAGS) on the stamped plan, "
                         "e.g. 3 = loads from L2 and no stores.  Outputs are then meaningless and not compared")
    args = ap.parse_args()
    shapes = [tuple(int(v) for v in sh.split(",")) for sh in (args.shape or ["3,1,8,256", "2,0,16,256"])]
    torch.cuda.set_device(0)
    n = 1 << args.logn
    dev = torch.device("cuda", 0)
    x = synth_batch(torch, args.batch, n, dev) if args.word_bytes == 8 else synth_u32(torch, args.batch, n, args.p, dev)
    y, yref = torch.empty_like(x), torch.empty_like(x)
    stream = torch.cuda.current_stream()
    LS = _lib.open_library(args.lib)
    LS.ntt_stamps_set.argtypes = [C.c_void_p, C.c_size_t]
    LP = _lib.open_library(os.path.join(ROOT, "ntt_aie_amd", "libntt_hip.so"))
    plans = {}
    for name, L in (("stamps", LS), ("product", LP)):
        h = C.c_void_p()
        assert L.ntt_plan_create(C.byref(h), args.logn, args.p, args.word_bytes, 0) == 0
        assert L.ntt_plan_generate_twiddles(h, 0, args.g) == 0
        plans[name] = (L, h)
    if args.dbg:
        LS.ntt_plan_set_debug.argtypes = [C.c_void_p, C.c_int]
        assert LS.ntt_plan_set_debug(plans["stamps"][1], args.dbg) == 0
    # the decomposition the launcher runs for THIS batch (plan alternatives): one record region per pass kind (CONTIG, column)
    alt = int(LS.ntt_plan_select(plans["stamps"][1], args.batch))
    npass = int(LS.ntt_plan_info(plans["stamps"][1], 256 + 16 * alt))
    assert npass in (1, 2) and npass == len(shapes), "one --shape per pass; at most one CONTIG and one column pass (one record region each)"
    stages = [int(LS.ntt_plan_info(plans["stamps"][1], 256 + 16 * alt + 1 + i)) for i in range(npass)]
    records = 1 << 18  # two regions of 2^17 wave records (a launch of the headline has 8192 x 4 / 16384 x 4 waves)
    buf = torch.zeros((records, REC), dtype=torch.int64, device="cuda:0")
    ms, k = (C.c_float * 8)(), C.c_int(0)

    def fwd(name, out, timed=False):
        L, h = plans[name]
        if timed:
            assert L.ntt_forward_profile(h, x.data_ptr(), out.data_ptr(), args.batch, 0, stream.cuda_stream, ms, 8, C.byref(k)) == 0
            return [float(ms[i]) for i in range(k.value)]
        assert L.ntt_forward(h, x.data_ptr(), out.data_ptr(), args.batch, 0, stream.cuda_stream) == 0
        return None

    for _ in range(6):
        fwd("product", yref)
        fwd("stamps", y)
    torch.cuda.synchronize()
    # per-pass kernel times, interleaved; here the stamps go to the dummy record (same store count per wave as with a buffer)
    tp, ts = [], []
    for _ in range(5):
        tp.append(fwd("product", yref, True))
        ts.append(fwd("stamps", y, True))
    same = bool(torch.equal(y, yref)) or bool(args.dbg)
    assert LS.ntt_stamps_set(buf.data_ptr(), records) == 0
    for _ in range(args.reps):
        fwd("stamps", y)
    torch.cuda.synchronize()
    same = same and (bool(torch.equal(y, yref)) or bool(args.dbg))
    assert LS.ntt_stamps_set(None, 0) == 0
    recs = buf.cpu().numpy().view(np.uint64)
    half = records // 2

    def med(rows, i):
        return statistics.median(r[i] for r in rows)

    regions = [recs[:half], recs[half:]]
    out = {"src_hash": _lib.kernel_source_hash(), "device": torch.cuda.get_device_name(0), "logn": args.logn, "batch": args.batch,
           "modulus": args.p, "word_bytes": args.word_bytes,
           "library": os.path.relpath(args.lib, ROOT), "outputs_identical_to_product_library": (None if args.dbg else same), "debug_flags": args.dbg,
           "method": __doc__.split("usage")[0].strip(),
           "overhead": {"product_pass_ms": [med(tp, i) for i in range(npass)], "stamped_pass_ms": [med(ts, i) for i in range(npass)],
                        "stamped_over_product": [med(ts, i) / med(tp, i) for i in range(npass)]},
           "passes": [dict(kind="%s pass, %d stages: %d register rounds of radix %d%s, %d threads per workgroup" % (
                               "CONTIG" if i == 0 else "column", stages[i], sh[0], sh[2], ", LDS-DMA tile" if sh[1] else "", sh[3]),
                           **reduce_region(regions[i], sh[0], bool(sh[1]), stages[i], sh[2], sh[3])) for i, sh in enumerate(shapes)]}
    if args.trace_prefix:
        for i, sh in enumerate(shapes):
            name = "contig" if i == 0 else "column"
            ev = chrome_trace(regions[i], sh[0], bool(sh[1]), args.trace_cycles, "%s pass, N=2^%d batch %d" % (name, args.logn, args.batch))
            with open("%s_%s.json" % (args.trace_prefix, name), "w") as f:
                json.dump(ev, f)
            out["passes"][i]["trace_file"] = os.path.basename("%s_%s.json" % (args.trace_prefix, name))
            out["passes"][i]["trace_events"] = len(ev)
    json.dump(out, sys.stdout, indent=1)
    print()
    if not same:
        sys.exit(1)


if __name__ == "__main__":
    main()
