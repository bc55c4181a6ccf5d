#!/usr/bin/env python3
"""bench.py -- forward-NTT throughput of the MI355X engine on BASELINE.json's headline
configuration: N = 2^16, Goldilocks prime 2^64-2^32+1, batch 4096 per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N --single-process ...      one process, N devices through ntt_plan_clone (SURVEY 8e's other form)

One step = one forward transform of the whole resident batch (input buffer -> output
buffer, both already in HBM).  Weak scaling: every rank owns `batch` polynomials and never
talks to the others in the timed region; the only collective is the one-off twiddle-table
broadcast from rank 0 (RCCL).  Rank 0 prints ONE JSON line.

No time without a check (the reference prints its launch times and then compares every word, src/test.cpp:203-247):
after the timed region EVERY rank (every device in --single-process) inverts its own output and compares it with its own
input, and checks out[b][0] == sum(a[b][:]) mod p on sampled rows (an oracle-free invariant of the network); the verdicts
are reduced with all_reduce(MIN), every rank's device identity and step time are gathered into the line, and the process
exits non-zero when any rank failed.

Every number of the `roofline` object is measured in THIS run on rank 0 -- per-pass kernel durations
(hipEvents on the launch stream), a device copy of the same bytes (the achievable stream rate beside
the 8 TB/s spec peak), the VALU floor (the same kernels of the experiment build with their loads
served from L2 and their stores skipped) -- except the hardware-counter figures (`traffic`, VALU
instruction counts), which need rocprofv3: those are read from profiles/ and quoted ONLY when the
kernel-source hash stamped into them equals the hash of the tree this run executes (null otherwise).
The CPU baseline is the oracle restatement of src/test.cpp:34-60 timed on a bounded sample.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GOLDILOCKS = 0xFFFFFFFF00000001
sys.path.insert(0, os.path.join(ROOT, "tools"))
from hw import (HBM_PEAK_GBS, PEAK_CLOCK_GHZ, SIMDS, VALU_PEAK_CYCLES_PLAIN, VALU_PEAK_CYCLES_VOP3, VALU_PEAK_SOURCE,  # noqa: E402
                valu_frac_of_peak, valu_peak_cycles)  # tools/hw.py: the one place the peaks live

SEED = 0x9E3779B97F4A7C15  # a[b][i] = splitmix64(SEED + b*N + i) mod p
PROFILE_ROUND = "r06"  # the collection DESIGN.md section 4 is generated from (tools/design_table.py); counters are quoted from the newest matching round (tagged_profile)


def _s64(v: int) -> int:
    """64-bit pattern as the signed value torch.int64 holds."""
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >> 63 else v


def splitmix64_words(torch, first_index: int, count: int, device):
    """splitmix64 of the words first_index .. first_index+count-1: z = index + 0x9E3779B97F4A7C15, then the generator's
    two xor-shift-multiply rounds and final xor-shift.  Returned as bit patterns in an int64 tensor (int64
    arithmetic wraps; logical right shifts are arithmetic shifts masked)."""
    z = torch.arange(count, dtype=torch.int64, device=device) + _s64(first_index + 0x9E3779B97F4A7C15)

    def lsr(v, k):
        return (v >> k) & ((1 << (64 - k)) - 1)

    z = (z ^ lsr(z, 30)) * _s64(0xBF58476D1CE4E5B9)
    z = (z ^ lsr(z, 27)) * _s64(0x94D049BB133111EB)
    return z ^ lsr(z, 31)


def synth_batch(torch, batch, n, device, seed=SEED, first_row=0):
    """[batch][n] canonical Goldilocks residues: a[b][i] = splitmix64(seed + (first_row + b)*n + i) mod p.
    A 64-bit word u is >= p only in [p, 2^64) = the signed range [-(2^32-1), -1], where u - p = u + 2^32 - 1."""
    out = torch.empty((batch, n), dtype=torch.int64, device=device)
    rows = max(1, (1 << 24) // n)  # 128 MiB of temporaries at a time
    for r0 in range(0, batch, rows):
        r1 = min(batch, r0 + rows)
        z = splitmix64_words(torch, seed + (first_row + r0) * n, (r1 - r0) * n, device)
        z = torch.where((z < 0) & (z >= -(2**32 - 1)), z + (2**32 - 1), z)
        out[r0:r1] = z.view(r1 - r0, n)
    return out


def host_cores():
    """(cores this process may run on, the cgroup CPU quota in cores or None)."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        pass
    return aff, quota


def cpu_baseline(logn, p, table, rows_fn, batch, cpu_seconds=20.0, threads=None):
    """Oracle (port of the reference CPU verification path, src/test.cpp:34-60: three `%` per butterfly) on the host cores,
    bounded sample: one thread (the reference is single-threaded) and ALL the cores this process may run on,
    one polynomial per task; the count is stated next to the figure.

    The sample is the GPU's own input and nothing else (the reference runs its CPU path on exactly the a[i] it handed the
    device, src/test.cpp:203-207): `rows_fn(k)` returns rows [0, k) of the resident device buffer copied back, k <= batch.  The
    sample is sized for about `cpu_seconds` of CPU work and CAPPED at the batch: a host fast enough to want more rows simply
    finishes sooner (`rows_beyond_gpu_batch` is 0 by construction and says so in the line)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O

    n = 1 << logn
    aff, quota = host_cores()
    # every core this process can actually run on: the affinity mask, capped by the cgroup CPU quota (on the GPU boxes the mask
    # shows all 256 host threads but the container is throttled to 16 cores: 256 OpenMP threads then run slower than 16)
    avail = aff if not quota else max(1, min(aff, int(quota + 0.999)))
    cores = int(threads) if threads else avail
    probe = rows_fn(min(2, batch))
    t0 = time.perf_counter()
    O.ntt(probe, table, p, nthreads=1)
    t1 = (time.perf_counter() - t0) / len(probe)
    rate_1 = 1.0 / t1
    # bounded sample: about `cpu_seconds` of CPU work in total, spread over the host threads (at least 4 polynomials per thread
    # when the batch has them), never more rows than the GPU transformed
    want = int(max(cores * 4, min(16384, cpu_seconds / t1)))
    sample = max(1, min(want, batch))
    a = rows_fn(sample)
    t0 = time.perf_counter()
    O.ntt(a, table, p, nthreads=cores)
    tn = time.perf_counter() - t0
    rate_n = sample / tn
    best, used = (rate_n, cores) if rate_n >= rate_1 else (rate_1, 1)
    return {"value": best, "unit": "NTT/s", "cores": used, "kind": "port",
            "sample": "rows 0..%d of the GPU's own input batch, N=2^%d, %d threads, %.2f s" % (sample - 1, logn, cores, tn),
            "sample_rows": sample, "sample_is_gpu_input": True, "rows_beyond_gpu_batch": 0,
            "sample_rows_wanted_for_%ds" % int(cpu_seconds): want, "sample_capped_at_batch": bool(want > batch),
            "host_affinity_cores": aff, "host_cgroup_quota_cores": quota, "host_available_cores": avail, "threads_all_cores_leg": cores,
            "value_all_cores": rate_n, "value_1thread": rate_1, "butterflies_per_s": best * (n // 2) * logn}


def device_copy_rate(torch, x, y, stream, reps=10, rounds=3):
    """The achievable stream rate of this device, in this process, on these buffers: read every byte of x once and
    write it once to y (the same algorithmic bytes as one transform).  Two forms -- the runtime's device-to-device copy and a
    plain elementwise kernel -- `rounds` timed bursts of `reps` each, the best burst kept: the figure is a capability (what
    plain streaming CAN reach here), so the fastest observation is the one that bounds the transform's trips."""
    def timed(fn):
        for _ in range(3):
            fn()
        best = None
        for _ in range(rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(reps):
                fn()
            e1.record(stream)
            e1.synchronize()
            ms = e0.elapsed_time(e1) / reps
            best = ms if best is None or ms < best else best
        return best

    nbytes = 2.0 * x.numel() * x.element_size()
    ms_copy = timed(lambda: y.copy_(x))
    ms_elem = timed(lambda: torch.bitwise_xor(x, 1, out=y))
    best = min(ms_copy, ms_elem)
    return {"GBs": nbytes / (best * 1e-3) / 1e9, "ms": best, "ms_memcpy_d2d": ms_copy, "ms_elementwise_kernel": ms_elem,
            "bytes": nbytes, "what": "best of %d bursts of %d back-to-back copies, two forms" % (rounds, reps)}


def valu_floor(torch, logn, p, batch, x, y, stream, reps=5):
    """The same pass kernels with every iteration's loads redirected to polynomial group 0 (L2-resident) and the
    stores skipped: what the butterflies + LDS exchanges cost with no HBM traffic.  Runs the experiment build
    (libntt_hip_exp.so = the same sources + -DNTT_EXPERIMENT; the product library has no such switch); outputs of
    these launches are meaningless and go to the scratch buffer y.  Returns (ms per pass | None, source / reason);
    the debug switch is set through the experiment library's own ntt_plan_set_debug(), not the process environment."""
    path = os.path.join(ROOT, "ntt_aie_amd", "libntt_hip_exp.so")
    if not os.path.exists(path):
        return None, "libntt_hip_exp.so absent (make -C ntt_aie_amd/csrc exp)"
    from ntt_aie_amd import _lib

    try:
        L = _lib.open_library(path)
        L.ntt_plan_set_debug.argtypes = [C.c_void_p, C.c_int]
    except (OSError, AttributeError) as e:
        return None, "libntt_hip_exp.so unusable: %s" % e
    h = C.c_void_p()
    rc = L.ntt_plan_create(C.byref(h), logn, p, 8, x.device.index)
    if rc != 0:
        return None, "experiment build: ntt_plan_create rc=%d" % rc
    try:
        rc = L.ntt_plan_set_debug(h, 3)
        if rc != 0:
            return None, "experiment build: ntt_plan_set_debug rc=%d" % rc
        rc = L.ntt_plan_generate_twiddles(h, 0, 7)
        if rc != 0:
            return None, "experiment build: ntt_plan_generate_twiddles rc=%d" % rc
        ms, k = (C.c_float * 8)(), C.c_int(0)
        best = None
        for _ in range(reps + 1):
            rc = L.ntt_forward_profile(h, x.data_ptr(), y.data_ptr(), batch, 0, stream.cuda_stream, ms, 8, C.byref(k))
            if rc != 0:
                return None, "experiment build: ntt_forward_profile rc=%d" % rc
            cur = [float(ms[i]) for i in range(k.value)]
            best = cur if best is None or sum(cur) < sum(best) else best
        return best, "measured in this run: libntt_hip_exp.so, ntt_plan_set_debug(3) = loads from L2, stores skipped"
    finally:
        L.ntt_plan_destroy(h)


def tagged_profile(name, src_hash):
    """The newest profiles/rNN_<name>.json that was collected on exactly these kernel sources, else (None, reason).
    Rounds are searched newest first (the current collection tag PROFILE_ROUND, then older ones): a round that did not touch
    the kernels keeps quoting the previous round's counters, a round that did quotes nothing until it has re-collected."""
    import glob
    import re

    cands = []
    for path in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_%s.json" % name)):
        m = re.match(r"r(\d\d)_", os.path.basename(path))
        if m and os.path.basename(path) == "r%s_%s.json" % (m.group(1), name):
            cands.append((int(m.group(1)), path))
    if not cands:
        return None, "profiles/rNN_%s.json absent" % name
    seen = []
    for _, path in sorted(cands, reverse=True):
        d = json.load(open(path))
        if d.get("src_hash") == src_hash:
            return d, "profiles/%s (src_hash %s)" % (os.path.basename(path), src_hash)
        seen.append("%s: %s" % (os.path.basename(path), d.get("src_hash")))
    return None, "no profiles/rNN_%s.json was collected on kernel sources %s (%s): not quoted" % (name, src_hash, "; ".join(seen))


def forward_counters(summary, passes, field="FieldGL"):
    """[entry per plan pass] from a tools/*_summary.py table, FORWARD kernels only (tools/kernel_key.py: an entry whose
    PassCfg INV argument is true can never be returned), or (None, reason)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from kernel_key import forward_entry

    out = []
    for kind, _, stages in passes:
        hit, why = forward_entry(summary["kernels"], kind == "contig", stages, field)
        if hit is None:
            return None, why
        out.append(hit)
    return out, None


def stream_plain_counts(passes):
    """(VALU instructions, of which plain moves / adds) of ONE forward Goldilocks butterfly statement per pass, from the generator's
    own instruction lists (tools/valu_mix.py -> tools/gen_gl_asm.py: no GPU, no file): the first pass reads its twiddles from VGPRs,
    column passes from SGPRs -- same counts, 22 VALU of which 2 moves."""
    from valu_mix import stream_mix

    out = []
    for kind, _, _ in passes:
        m = stream_mix("fwd", kind != "contig")
        out.append((m["valu"], m["mix"].get("plain", 0)))
    return out


def stream_source_hash():
    """identity of what tools/stream_occupancy measures: the statement generator and the probe (16 hex digits)"""
    import hashlib

    h = hashlib.sha256()
    for f in ("gen_gl_asm.py", "stream_occupancy.hip"):
        h.update(open(os.path.join(ROOT, "tools", f), "rb").read())
    return h.hexdigest()[:16]


def statement_steady_state():
    """{waves per SIMD: cycles per butterfly per SIMD} of the forward butterfly statement ALONE in steady state (many generations of
    workgroups: tools/stream_occupancy.hip), from the newest profiles/rNN_stream_occupancy.txt whose `# stream_src_hash` line equals
    stream_source_hash() of this tree (tools/run_round.sh writes it), or None: a statement-alone figure that was not re-measured on
    this tree's generator is not quoted."""
    import glob

    want = stream_source_hash()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_stream_occupancy.txt")), reverse=True):
        rows, on, stamped = {}, False, None
        for line in open(path):
            if line.startswith("# stream_src_hash"):
                stamped = line.split()[-1]
            elif line.startswith("# steady state"):
                on = True
            elif on and line[:1].isdigit():
                f = line.split()
                rows[int(f[0])] = float(f[1])
        if stamped == want and rows:
            return {"cycles_per_butterfly_by_waves_per_simd": rows, "source": os.path.relpath(path, ROOT)}
    return None


def valu_roofline(sq_entries, passes, per_pass_ms, batch, logn, stream_counts=None, statement=None):
    """The vector-ALU roofline of the forward transform on the unit's MEASURED throughput (tools/hw.py, profiles/r06_valu_peak.txt:
    4 cycles per wave64 instruction for the VOP3-class forms the butterfly statements are made of -- SGPR-pair carries, 64-bit compare,
    select by SGPR pair, v_mad_u64_u32 -- 2 cycles for plain moves and adds):
      peak cycles per butterfly = stream's VOP3-class instructions x 4 + its moves x 2 + (SQ_INSTS_VALU per butterfly - stream length) x 4
      frac_of_peak_at_held_clock = peak cycles per butterfly x wave-butterflies / (SIMDs x the launch's shader cycles): clock-free
    stream_counts = [(VALU per butterfly of the statement, of which plain)] per pass (stream_plain_counts); None prices EVERY counted
    instruction at 4 cycles (an upper estimate).  statement = statement_steady_state(): what the statement alone sustains."""
    n = 1 << logn
    bf = [batch * (n // 2) * stages for _, _, stages in passes]            # butterflies per launch of each pass
    ipb = [e[1]["valu_instr_per_butterfly"] for e in sq_entries]          # SQ_INSTS_VALU / wave-butterflies, forward kernels
    mean_ipb = sum(i * b for i, b in zip(ipb, bf)) / sum(bf)
    held, held_note = sane_clocks([e[1].get("held_clock_GHz") for e in sq_entries])
    cyc = [e[1].get("kernel_cycles") for e in sq_entries]
    waves = [e[1].get("mean_waves_per_simd") for e in sq_entries]
    stall = [e[1].get("wave_issue_stall_frac") for e in sq_entries]
    achieved = sum(bf) / (sum(per_pass_ms) * 1e-3)
    plain = [min(i, sc[1]) for i, sc in zip(ipb, stream_counts)] if stream_counts else [0.0] * len(ipb)
    peak_cyc = [valu_peak_cycles(i, pl) for i, pl in zip(ipb, plain)]     # SIMD cycles one wave-butterfly needs at the unit's throughput
    mean_peak_cyc = sum(c * b for c, b in zip(peak_cyc, bf)) / sum(bf)

    def peak(f_ghz, cycles_per_bf):
        return SIMDS * f_ghz * 1e9 / cycles_per_bf * 64

    out = {
        "instr_per_butterfly": ipb, "instr_per_butterfly_mean": mean_ipb, "plain_instr_per_butterfly": plain,
        "peak_cycles_per_wave_instr": {"vop3_class": VALU_PEAK_CYCLES_VOP3, "plain_moves_adds": VALU_PEAK_CYCLES_PLAIN, "source": VALU_PEAK_SOURCE},
        "peak_cycles_per_butterfly": peak_cyc,
        "peak_butterflies_per_s": peak(PEAK_CLOCK_GHZ, mean_peak_cyc), "peak_clock_GHz": PEAK_CLOCK_GHZ,
        "achieved_butterflies_per_s": achieved,
        "frac_of_peak_at_2.4GHz": achieved / peak(PEAK_CLOCK_GHZ, mean_peak_cyc),
        "frac_of_peak_at_2.4GHz_per_pass": [b / (t * 1e-3) / peak(PEAK_CLOCK_GHZ, c) for b, t, c in zip(bf, per_pass_ms, peak_cyc)],
        "kernels": [e[0] for e in sq_entries],
        "mean_waves_per_simd": waves, "wave_issue_stall_frac": stall,
    }
    if held_note:
        out["held_clock_GHz"], out["held_clock_note"] = held, held_note
    if all(h for h in held) and all(c for c in cyc):
        out["held_clock_GHz"] = held
        kcyc = [SIMDS * c / (b / 64) for b, c in zip(bf, cyc)]              # kernel cycles per wave-butterfly per SIMD
        per_pass = [p / k for p, k in zip(peak_cyc, kcyc)]
        tot_cyc = sum(cyc)
        out["kernel_cycles_per_wave_butterfly_per_simd"] = kcyc
        out["frac_of_peak_at_held_clock_per_pass"] = per_pass
        out["frac_of_peak_at_held_clock"] = sum(p * b / 64 for p, b in zip(peak_cyc, bf)) / (SIMDS * tot_cyc)
        # the clock THIS run held, if a launch takes the same number of cycles as under the profiler
        out["clock_this_run_GHz_estimate"] = [c / (t * 1e6) for c, t in zip(cyc, per_pass_ms)]
        if statement:
            st = statement["cycles_per_butterfly_by_waves_per_simd"]
            best = min(v for w, v in st.items() if w >= 4) if any(w >= 4 for w in st) else min(st.values())
            out["statement_alone_steady_state"] = {
                "cycles_per_butterfly_by_waves_per_simd": st, "cycles_per_butterfly_at_4_or_more_waves": best, "source": statement["source"],
                "kernel_over_statement": [k / best for k in kcyc]}  # (what the rounds could run at if exchanges, loads and stores cost nothing)
        out["saturated"] = bool(out["frac_of_peak_at_held_clock"] >= SATURATED)
    return out


# A roofline is NAMED as the bound only at this fraction of it.  The headline's passes stream at 0.87-0.93 of a same-run device copy
# from box to box (the copy's own rate moves 3 % between runs): a 0.9 threshold made the label flip between two runs of one build.
SATURATED = 0.95

# A held clock is GRBM_GUI_ACTIVE / 8 / the launch's duration.  For launches of a few microseconds the counter also sees the
# command processor's work around the kernel, and the quotient comes out ABOVE the part's peak clock (round 5's line carried 3.30
# GHz for config 2's 14 us launch): such a figure is not a clock.  It is nulled where the valu object is built, so the printed
# line, decide_bound and DESIGN's table all see the same sanitised value.
CLOCK_SANITY = 1.02  # tolerance on PEAK_CLOCK_GHZ


def sane_clocks(clocks):
    """(clocks with impossible entries nulled, reason or None)"""
    out = [h if (h and h <= PEAK_CLOCK_GHZ * CLOCK_SANITY) else None for h in clocks]
    bad = any(h and h > PEAK_CLOCK_GHZ * CLOCK_SANITY for h in clocks)
    return out, ("quotient above the %.1f GHz peak clock: the launch is too short for GRBM_GUI_ACTIVE / 8 to be the kernel's own cycles"
                 % PEAK_CLOCK_GHZ) if bad else None


# ---- the printed line: numbers, not commentary (the reference prints numbers, src/test.cpp:171-174) -------------------------
# What every key means is DESIGN.md section 4 ("Reading the line"), not a string in the line.  The few keys that still explain other
# keys are dropped from the default line, provenance strings are cut to the file they name and nested floats carry 5 significant
# digits, so that the driver-run line stays below 6 KB; `--explain` prints the full form (provenance in full, full precision:
# tools/design_table.py and the counter-provenance tests read that).
PROSE_KEYS = {"what", "definition", "bound_note", "frac_of_practical_hbm_what", "verdict", "data_note"}
TOP_LEVEL_EXACT = {"value", "ms_per_step", "butterflies_per_s", "ops_per_s_reference_convention"}  # (the last two: 7 digits, below)


# per extra configuration (out["configs"][i]) the slim line keeps the measurement and the verdicts; the shape's description is
# tools/configs.py's (by `key`), the generator is the headline's
CONFIG_DROP = {"name", "data", "logn", "word_bytes", "modulus", "batch", "op",  # the shape is tools/configs.py's, by `key`
               "ms_back_to_back"}                                                 # == ms; every MEASURED key stays
CONFIG_ROOFLINE_DROP = {"bound_detail", "roofline_of_fields"}  # (the numbers the detail is made of are keys of their own)
CONFIG_VALU_DROP = {"kernels", "kernel_cycles", "held_clock_note"}


def slim_line(out):
    """the default line from the full one: no prose keys, short provenance, 5 significant digits below the top level"""
    import re

    def short(k, v):
        if (k.endswith("_source") or k == "source") and isinstance(v, str):
            for cut in (";", " -- ", " ("):
                v = v.split(cut)[0]
            return v[9:] if v.startswith("profiles/") else v[:96]  # a bare rNN_*.json / .txt is a file under profiles/
        if k == "kernels" and isinstance(v, list):  # a kernel's identity = its PassCfg<...> argument list (tools/kernel_key.py)
            def ident(x):
                m = re.search(r"PassCfg<([^<>]*)>", x)
                return (m.group(1) if m else x[:64]).replace("ntt::", "").replace(" ", "")
            return [ident(x) if isinstance(x, str) else x for x in v]
        if k == "statement_alone_steady_state" and isinstance(v, dict):
            return {q: v[q] for q in ("cycles_per_butterfly_at_4_or_more_waves", "kernel_over_statement", "source") if q in v}
        return v

    def walk(o, top=False):
        if isinstance(o, dict):
            return {k: (v if (top and k in TOP_LEVEL_EXACT) else walk(short(k, v))) for k, v in o.items() if k not in PROSE_KEYS}
        if isinstance(o, list):
            return [walk(v) for v in o]
        if isinstance(o, float):
            v = float("%.5g" % o)
            return int(v) if abs(v) >= 1e6 and v == int(v) else v  # (4295000000 instead of 4295000000.0)
        return o

    slim = walk(out, top=True)
    for k in ("butterflies_per_s", "ops_per_s_reference_convention"):  # derived from `value`: 7 digits
        if isinstance(slim.get(k), float):
            slim[k] = float("%.7g" % slim[k])
    for e in slim.get("configs") or []:
        for k in CONFIG_DROP:
            e.pop(k, None)
        r = e.get("roofline") or {}
        for k in CONFIG_ROOFLINE_DROP:
            r.pop(k, None)
        if isinstance(r.get("valu"), dict):
            r["valu"] = {k: v for k, v in r["valu"].items() if k not in CONFIG_VALU_DROP}
        if e.get("key") == "cfg5_shard" and "headline" in str((out.get("configs") or [{}] * 9)[slim["configs"].index(e)].get("roofline", {}).get("valu_source", "")):
            # its counters are the HEADLINE's (the same two kernels at twice the rows, scaled): the numbers are in `roofline` above
            r["valu"], r["valu_source"], r["traffic_source"] = None, "= roofline.valu (the headline's kernels)", "= roofline.traffic x rows / 4096"
    h = slim.get("roofline") or {}
    if isinstance(h.get("valu"), dict):
        # strings and constants: the kernels' identities are the keys of the file valu_source names, the prices are tools/hw.py's,
        # the statement's plain-instruction count is the generator's (22 VALU of which 2 moves)
        for k in ("kernels", "held_clock_note", "peak_cycles_per_wave_instr", "peak_clock_GHz", "plain_instr_per_butterfly"):
            h["valu"].pop(k, None)
    for r in slim.get("ranks") or []:
        if r.get("pci_bus_id"):  # one identity per rank is enough in the slim line
            r.pop("uuid", None)
            r.pop("name", None)
    c = slim.get("cpu_baseline") or {}
    for k in [k for k in c if k.startswith("sample_rows_wanted") or k in ("sample_capped_at_batch", "threads_all_cores_leg")]:
        c.pop(k)
    if isinstance(h.get("valu_floor_source"), str):
        h["valu_floor_source"] = h["valu_floor_source"].split(":")[0]
    return slim


def emit(out, args):
    print(json.dumps(out if getattr(args, "explain", False) else slim_line(out), separators=(",", ":")), flush=True)


def decide_bound(pass_frac_of_copy, valu_frac_of_peak, waves=None, held_clock_GHz=None, weights=None):
    """roofline.bound from the run's own numbers, never asserted:
      "hbm"        every pass streams at >= SATURATED (0.95) of the same-run device copy
      "valu"       the vector ALU is at >= 0.95 of its measured throughput (tools/hw.py) at the held clock
      "power-cap"  neither, and the kernels hold -- on the TIME-WEIGHTED mean over the operation's kernels (`weights`: their
                   durations or cycles; plain mean without) -- less than 0.9 of the 2.4 GHz peak clock: one short kernel a few
                   per cent under the threshold does not decide the label.  The clock is the COUNTER run's (serialised rocprofv3
                   --pmc launches), said so in the detail; DESIGN.md section 4 has the power-probe evidence behind the name.
      "unsaturated" none of the above can be shown (no counters for these sources, or the clock is held): nothing is claimed
    Returns (bound, detail).  Pure arithmetic (CPU unit test)."""
    hb = min(pass_frac_of_copy) if pass_frac_of_copy else None
    if hb is not None and hb >= SATURATED:
        return "hbm", "every pass streams at >= %.2f of the same-run device copy" % hb
    if valu_frac_of_peak is not None and valu_frac_of_peak >= SATURATED:
        return "valu", "vector ALU at %.2f of its measured throughput at the held clock" % valu_frac_of_peak
    w = ("%.1f" % (sum(waves) / len(waves))) if waves and all(waves) else "~4"
    what = "copy %s, valu %s, %s waves per SIMD" % (
        "%.2f" % hb if hb is not None else "n/a", "%.2f" % valu_frac_of_peak if valu_frac_of_peak is not None else "n/a (no counters)", w)
    clk = None
    if held_clock_GHz and all(held_clock_GHz):
        ws = list(weights) if weights and len(weights) == len(held_clock_GHz) and all(weights) else [1.0] * len(held_clock_GHz)
        clk = sum(h * x for h, x in zip(held_clock_GHz, ws)) / sum(ws)
    if clk is not None and clk < 0.9 * PEAK_CLOCK_GHZ:
        return "power-cap", "neither unit saturated (%s); clock %.2f of %.1f GHz (counter run, time-weighted; lowest %.2f)" % (
            what, clk, PEAK_CLOCK_GHZ, min(held_clock_GHz))
    return "unsaturated", "neither unit saturated (%s); no held-clock figure shows a power cap" % what


# ---- verification of a shard: every rank, every device --------------------------------------------------------------------
def rowsum_mod_p(rows, p):
    """[sum(row) mod p] for uint64 rows: 32-bit halves summed in 64 bits (N <= 2^28 words cannot overflow), combined as
    Python integers.  The network's output 0 is the plain coefficient sum (every stage adds the pair into the lower index)."""
    import numpy as np

    rows = np.ascontiguousarray(rows).view(np.uint64)
    lo = (rows & np.uint64(0xFFFFFFFF)).sum(axis=1, dtype=np.uint64)
    hi = (rows >> np.uint64(32)).sum(axis=1, dtype=np.uint64)
    return [((int(h) << 32) + int(l)) % p for h, l in zip(hi, lo)]


def verify_shard(torch, plan, x, y, p, stream, sample=16):
    """Called right after the timed region, y = forward(x) still untouched.  (1) inverse(y) == x, whole shard, word for word;
    (2) y[b][0] == sum(x[b][:]) mod p and y[b][0] < p on `sample` rows spread over the shard -- an invariant of the forward
    network that needs no oracle.  The reference's rule: no time is reported without a full compare (src/test.cpp:203-247)."""
    import numpy as np

    batch = x.shape[0]
    back = plan.inverse(y, stream=stream)
    stream.synchronize()
    rt = bool(torch.equal(back, x))
    del back
    rows = sorted(set(int(r) for r in np.linspace(0, batch - 1, min(sample, batch))))
    idx = torch.tensor(rows, device=x.device)
    xs = x.index_select(0, idx).cpu().numpy().view(np.uint64)
    y0 = y[:, 0].index_select(0, idx).cpu().numpy().view(np.uint64)
    got = [int(v) for v in y0]
    sums = got == rowsum_mod_p(xs, p) and all(v < p for v in got)
    return {"round_trip_identical": rt, "coefficient_sum_invariant": bool(sums), "rows_sampled": len(rows)}


def device_identity(torch, index):
    """What tells two GPUs apart in the line: PCI bus id (HIP runtime, through the copy of libamdhip64 torch already loaded)
    and the device UUID / name torch reports."""
    out = {"local_device": int(index), "pci_bus_id": None, "uuid": None, "name": None}
    try:
        props = torch.cuda.get_device_properties(index)
        out["name"] = props.name
        u = getattr(props, "uuid", None)
        out["uuid"] = str(u) if u is not None else None
        if hasattr(props, "pci_bus_id"):
            out["pci_bus_id"] = "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), props.pci_bus_id, getattr(props, "pci_device_id", 0))
    except Exception as e:  # identity is a report, never a reason to fail the run
        out["identity_error"] = repr(e)
    if out["pci_bus_id"] is None:
        try:
            path = next((l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64.so" in l), None)
            if path:
                hip = C.CDLL(path)
                buf = C.create_string_buffer(64)
                if hip.hipDeviceGetPCIBusId(buf, 64, int(index)) == 0:
                    out["pci_bus_id"] = buf.value.decode()
        except Exception as e:
            out["identity_error"] = repr(e)
    return out


def reduce_verdicts(dist, torch, dev, world, rank, flags, ident):
    """Every rank contributes its verdict flags (list of bool) and its identity record; returns (flags reduced with MIN over
    ranks, [record of rank 0, 1, ...]) on EVERY rank.  world == 1: no collective."""
    if world == 1:
        return [bool(f) for f in flags], [ident]
    t = torch.tensor([1 if f else 0 for f in flags], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    recs = [None] * world
    dist.all_gather_object(recs, ident)
    return [bool(v) for v in t.tolist()], recs


def verdict_fields(reduced, recs, world_seen, one_device_ok=False):
    """The line's verification block and the process exit code (non-zero when ANY rank failed either check, or when the ranks
    of a multi-rank job are not provably on `world_seen` distinct devices: a job whose ranks all landed on one GPU would
    otherwise print a green aggregate of one GPU's time-sliced work.  one_device_ok (the NTT_BENCH_ONE_DEVICE=1 rehearsal on a
    one-GPU box) waives the distinctness rule and says so in the line)."""
    ids = [r.get("pci_bus_id") or r.get("uuid") for r in recs]
    distinct = len(set(i for i in ids if i)) if any(ids) else None
    if world_seen <= 1 or one_device_ok:
        placed = True
    else:
        placed = distinct is not None and all(ids) and distinct == world_seen and len(recs) == world_seen
    ok = all(reduced) and placed
    return {"all_ranks_verified": ok, "world_size_seen": int(world_seen),
            "verification": {"round_trip_identical_all": reduced[0], "coefficient_sum_invariant_all": reduced[1],
                             "ranks_on_distinct_devices": bool(placed) if not one_device_ok else None,
                             "distinctness_waived_one_device_rehearsal": bool(one_device_ok),
                             "what": "every rank: inverse(forward(x)) == x over its whole shard (torch.equal) and out[b][0] == "
                                     "sum(a[b][:]) mod p on sampled rows; flags reduced with all_reduce(MIN); a multi-rank line is "
                                     "verified only when every rank reports a device identity and all of them differ"},
            "distinct_devices": distinct,
            "ranks": recs}, (0 if ok else 1)


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: the parent -- before anything touches the GPU -- starts the N ranks itself,
    one child process per rank with RANK / LOCAL_RANK / WORLD_SIZE in its environment and a FILE rendezvous in a fresh temporary
    directory (--rdzv-file: torch.distributed's FileStore; RCCL's own bootstrap sockets bind port 0 themselves).  No TCP port is
    probed, released and handed over, so there is nothing to collide on and nothing to retry: a failed launch is reported with
    its exit code, as the reference reports a failed run (src/test.cpp:162-166).  Rank 0's ONE JSON line is relayed.
    The torchrun form keeps working: with WORLD_SIZE set this function is never reached."""
    import subprocess
    import tempfile

    rehearsal = os.environ.get("NTT_BENCH_ONE_DEVICE") == "1"
    if not rehearsal:
        import torch  # device_count() does not initialise the GPU on this image

        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.stderr.write("bench.py: --gpus %d but only %d device(s) visible; refusing to oversubscribe a GPU "
                             "(NTT_BENCH_ONE_DEVICE=1 NTT_BENCH_BACKEND=gloo rehearses the launch path on one device)\n"
                             % (args.gpus, have))
            return 2
    base = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("MASTER_ADDR", "MASTER_PORT"):
        base.pop(k, None)
    with tempfile.TemporaryDirectory(prefix="ntt_rdzv_") as tmp:
        rdzv = os.path.join(tmp, "store")
        procs, logs = [], []
        for r in range(args.gpus):  # every rank's output goes to a file of its own: no pipe can fill up while another rank is awaited
            env = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus))
            fo, fe = open(os.path.join(tmp, "out%d" % r), "w+"), open(os.path.join(tmp, "err%d" % r), "w+")
            logs.append((fo, fe))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv + ["--rdzv-file", rdzv], env=env,
                                          stdout=fo, stderr=fe, text=True))
        # a rank that dies (before or after the rendezvous) would leave the others waiting on it: the first non-zero exit ends
        # the job -- the exact children started above are terminated, nothing is restarted.  A rank that HANGS instead of exiting
        # is ended the same way when the job's wall-clock limit expires (NTT_BENCH_LAUNCH_TIMEOUT_S, default 1500 s: a default
        # run takes a few minutes), and the parent exits 124 like timeout(1).
        deadline = time.monotonic() + float(os.environ.get("NTT_BENCH_LAUNCH_TIMEOUT_S", "1500"))
        timed_out = False
        while any(p.poll() is None for p in procs):
            timed_out = time.monotonic() > deadline
            if timed_out or any(p.poll() not in (None, 0) for p in procs):
                if timed_out:
                    sys.stderr.write("bench.py: the %d-rank job did not finish within its wall-clock limit; terminating the ranks still "
                                     "running: %s\n" % (args.gpus, [r for r, p in enumerate(procs) if p.poll() is None]))
                for p in procs:
                    if p.poll() is None:
                        p.terminate()
                for p in procs:
                    try:
                        p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                break
            time.sleep(0.05)
        codes = [p.wait() for p in procs]
        outs = []
        for fo, fe in logs:
            fo.seek(0)
            fe.seek(0)
            outs.append((fo.read(), fe.read()))
            fo.close()
            fe.close()
    for r, (_, err) in enumerate(outs):
        if err and (codes[r] != 0 or r == 0):
            sys.stderr.write(err if r == 0 else "[rank %d] %s" % (r, err))
    sys.stdout.write(outs[0][0] or "")
    sys.stdout.flush()
    bad = [c for c in codes if c > 0] or [1 for c in codes if c != 0]  # a rank's own exit code before "terminated by the parent"
    if timed_out:
        return 124
    return bad[0] if bad else 0


def config_name(logn, batch, world):
    if logn == 16 and batch == 4096:
        return "BASELINE config 3 forward leg (headline)" + (", weak-scaled" if world > 1 else "")
    if logn == 16 and batch == 8192:
        return "BASELINE config 5 (65536 rows over 8 GPUs = 8192 per GPU), %d GPU(s) here" % world
    return "off-headline shape (N=2^%d, %d per GPU)" % (logn, batch)


def base_line(args, logn, batch, world, value, elapsed, passes, table_broadcast, launch):
    """The contract's keys; `value` = whole-job NTT/s, `elapsed` = the timed region in seconds (max over ranks / devices)."""
    from ntt_aie_amd import _lib

    n = 1 << logn
    return {
        "metric": "forward-NTT/s, N=2^%d 64-bit Goldilocks prime, batch=%d per GPU" % (logn, batch),
        "value": value, "unit": "NTT/s", "butterflies_per_s": value * (n // 2) * logn,
        # the reference's own operation count (profile/plot_efficiency.py:25,44: 5.5 * N * log2 N per transform)
        "ops_per_s_reference_convention": value * 5.5 * n * logn,
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_steps": PREWARM,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic", "launch": launch,
        "config": {"workload": "%s: forward NTT N=2^%d, Goldilocks, make_roots(g=7) table, batch %d/GPU (%d in the job), out of place, "
                               "HBM-resident" % (config_name(logn, batch, world), logn, batch, batch * world),
                   "data_note": "a[b][i] = splitmix64(0x9E3779B97F4A7C15 + b*N + i) mod p (SURVEY 8d)",
                   "baseline_config": (3 if (logn == 16 and batch == 4096) else 5 if (logn == 16 and batch == 8192) else None),
                   "batch_per_gpu": batch, "hbm_passes": len(passes),
                   "sharding": "contiguous batch rows per rank, no data-path collective",
                   "table_broadcast": table_broadcast,
                   "kernel_src_hash": _lib.kernel_source_hash()},
    }


PREWARM = 8  # untimed, before the W warm-up steps: first touches of 4 GiB (TLB) and the clock ramp


def rank0_extras(torch, args, plan, table, x, y, stream, passes, out, world):
    """Rank 0 / device 0 only, AFTER the timed region and the verification: step-time distribution, the roofline object
    (per-pass hipEvents, device copy, VALU floor, counters quoted from profiles/ when their hash matches), the CPU baseline
    (N = 1 only), config 3's inverse leg and BASELINE configs 2 and 4 (N = 1, default shape).  Overwrites y.  Returns an exit
    code: non-zero when one of the extra configurations computed a wrong result."""
    import numpy as np

    from ntt_aie_amd import _lib

    logn, batch, p = args.logn, args.batch, GOLDILOCKS
    n = 1 << logn
    src_hash = _lib.kernel_source_hash()
    # per-step distribution (median and min): one hipEvent pair per step on the launch stream
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(max(args.steps, 5))]
    for _ in range(PREWARM):  # the verification above left the GPU idle: the same untimed lead-in as the timed region had
        plan.forward(x, y, stream=stream)
    for e0, e1 in evs:
        e0.record(stream)
        plan.forward(x, y, stream=stream)
        e1.record(stream)
    stream.synchronize()
    step_ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    out["step_ms_median"] = step_ms[len(step_ms) // 2]
    out["step_ms_min"] = step_ms[0]
    # roofline: per-pass kernel durations from hipEvents on the launch stream
    # (ntt_forward_profile blocks until its last event: two plain transforms are queued ahead of every profiled one, so the
    # profiled passes run behind a busy GPU at the clock the timed region held, not on a chip ramping up from idle)
    reps = max(5, min(args.steps, 20))
    per_pass = np.zeros(len(passes))
    for _ in range(reps):
        plan.forward(x, y, stream=stream)
        plan.forward(x, y, stream=stream)
        per_pass += np.array(plan.forward_profile(x, y, stream=stream))
    per_pass /= reps
    alg_bytes = 2.0 * n * 8 * batch  # 2*N*sizeof(word) per transform, read once + write once
    t_kernels = float(per_pass.sum()) * 1e-3
    achieved = alg_bytes / t_kernels / 1e9
    dom = int(per_pass.argmax())
    headline = logn == 16 and batch == 4096
    # measured here: what a plain copy of the same bytes achieves, and the VALU floor of the same kernels
    copy = device_copy_rate(torch, x, y, stream)
    floor, floor_src = (None, "skipped (--no-valu-floor)") if args.no_valu_floor else valu_floor(torch, logn, p, batch, x, y, stream)
    if floor is not None and len(floor) != len(per_pass):
        floor, floor_src = None, "experiment build ran %d passes, the product %d" % (len(floor), len(per_pass))
    # counters (rocprofv3 --pmc, separate runs of this command): quoted only when collected on these sources AND
    # only from entries of FORWARD kernels (the stored PassCfg<...> argument list says INV = false)
    traffic, traffic_src = None, "counters are collected for the headline configuration only"
    valu, valu_src = None, traffic_src
    if headline:
        d, traffic_src = tagged_profile("pmc_traffic", src_hash)
        if d:
            ent, why = forward_counters(d, passes)
            if ent:
                traffic = sum(e[1]["hbm_bytes_per_launch"] for e in ent)
                traffic_src += "; forward kernels: " + " + ".join(e[0] for e in ent)
            else:
                traffic_src += "; not quoted: " + why
        d, valu_src = tagged_profile("sq_counters", src_hash)
        if d:
            ent, why = forward_counters(d, passes)
            if ent:
                valu = valu_roofline(ent, passes, [float(v) for v in per_pass], batch, logn, stream_counts=stream_plain_counts(passes),
                                     statement=statement_steady_state())
            else:
                valu_src += "; not quoted: " + why
    step_s = out["ms_per_step"] * 1e-3
    pass_of_copy = [alg_bytes / (float(v) * 1e-3) / 1e9 / copy["GBs"] for v in per_pass]
    vfrac = valu.get("frac_of_peak_at_held_clock") if valu else None
    # the clock the kernels hold: GRBM_GUI_ACTIVE / 8 / duration of the profiled launches (a measurement; clock_this_run_GHz_estimate,
    # the same cycles over THIS run's durations, is reported beside it)
    bound, bound_detail = decide_bound(pass_of_copy, vfrac, valu.get("mean_waves_per_simd") if valu else None,
                                       valu.get("held_clock_GHz") if valu else None, [float(v) for v in per_pass])
    # what this pass count can reach on THIS device: every trip at the rate a plain copy of the same bytes achieves here
    practical_ms = len(passes) * copy["ms"]
    out["roofline"] = {
        # The contract's figure: algorithmic HBM bytes / kernel time against the 8 TB/s spec peak (achieved, peak, unit, frac);
        # `roofline_of_fields` says which roofline those four numbers are.  `bound` is decided from the run's numbers
        # (decide_bound): "hbm" / "valu" only when that unit is at >= 0.95 of what it can do; "power-cap" when neither is and the
        # kernels hold less than 0.9 of the peak clock; "unsaturated" when nothing can be shown.
        "bound": bound, "bound_detail": bound_detail, "roofline_of_fields": "hbm",
        "frac_ceiling": 1.0 / len(passes),
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        # the same bytes over the step time the line's own `value` is made of (launch gaps included), NOT clamped: a step cannot be
        # shorter than its kernels, so frac_step <= frac up to the clock the chip held in each phase (the two are measured seconds
        # apart); the GPU contract test asserts frac_step <= 1.03 x frac instead of hiding a violation behind a min()
        "frac_step": alg_bytes / step_s / 1e9 / HBM_PEAK_GBS,
        "achieved_step": alg_bytes / step_s / 1e9,
        "practical_hbm_floor_ms": practical_ms,
        "frac_of_practical_hbm": practical_ms / (step_s * 1e3),
        "traffic": traffic, "traffic_source": traffic_src,
        "algorithmic_bytes_per_transform": 2 * n * 8, "algorithmic_bytes_per_launch": alg_bytes,
        "passes": len(passes), "pass_stages": [stages for _, _, stages in passes],
        "pass_ms": [float(v) for v in per_pass], "dominant_pass": dom,
        # each pass kernel reads and writes every coefficient once: its own stream rate
        "pass_stream_GBs": [alg_bytes / (float(v) * 1e-3) / 1e9 for v in per_pass],
        "pass_stream_frac": [alg_bytes / (float(v) * 1e-3) / 1e9 / HBM_PEAK_GBS for v in per_pass],
        # the same bytes through a plain copy, same process, same buffers: the achievable rate beside the spec peak
        "device_copy": copy, "frac_of_device_copy": achieved / copy["GBs"],
        "pass_stream_frac_of_device_copy": pass_of_copy,
        # floor = the same kernels, loads from L2, no stores: what the butterflies + exchanges cost with no HBM traffic
        "valu_floor_pass_ms": floor,
        "valu_floor_frac_of_pass": ([f / float(v) for f, v in zip(floor, per_pass)] if floor else None),
        "valu_floor_source": floor_src,
        "valu": valu, "valu_source": valu_src,
    }
    if world == 1 and not args.no_cpu_baseline:
        def rows_fn(k):  # rows [0, k) of the resident input buffer, k <= batch (cpu_baseline caps its sample there)
            assert k <= batch
            return np.ascontiguousarray(x[:k].cpu().numpy().view(np.uint64))

        out["cpu_baseline"] = cpu_baseline(logn, p, table, rows_fn, batch, threads=args.cpu_threads or None)
    # BASELINE config 3 is forward + inverse: the inverse transform of the same batch, outside the timed region above
    # (scaled by N^-1, natural order in and out), one event pair per step; measured LAST so that nothing it allocates or heats
    # perturbs the roofline measurements above
    if not args.no_inverse:
        x2 = torch.empty_like(x)
        plan.forward(x, y, stream=stream)
        for _ in range(PREWARM + args.warmup):  # the same untimed lead-in as the forward leg (the CPU baseline above left the GPU idle)
            plan.inverse(y, x2, stream=stream)
        for e0, e1 in evs:
            e0.record(stream)
            plan.inverse(y, x2, stream=stream)
            e1.record(stream)
        stream.synchronize()
        inv_ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
        out["inverse"] = {"ms_per_step_median": inv_ms[len(inv_ms) // 2], "ms_per_step_min": inv_ms[0],
                          "NTT_per_s": batch / (inv_ms[len(inv_ms) // 2] * 1e-3),
                          "vs_forward_median": inv_ms[len(inv_ms) // 2] / out["step_ms_median"],
                          "round_trip_identical": bool(torch.equal(x2, x))}
        del x2
    # BASELINE configs 2 and 4 in the same driver-run line (one GPU, default shape only; --no-configs for counter collections,
    # whose per-kernel means must see the headline's launches alone).  A config whose result is WRONG turns the exit code red.
    code = 0
    if world == 1 and headline and not args.no_configs:
        out["configs"] = extra_configs(torch, stream, src_hash, steps=max(5, min(args.steps, 20)))
        out["configs_all_verified"], code = configs_verdict(out["configs"])
    return code


# ---- the other single-GPU BASELINE configurations, in the same driver-run line ---------------------------------------------
# The reference prints every timing it publishes from the one program the user runs (src/test.cpp:157-175).  After the headline's
# timed region and verification, rank 0 of a one-GPU default run measures BASELINE config 2 (N = 2^12, 32-bit prime, batch 1024,
# forward), config 4 (N = 2^20 negacyclic product, Goldilocks, batch 512) and the per-GPU shard of config 5 (8192 of the 8-GPU job's
# 65536 rows) for a few steps each and verifies each without the oracle; config 3's inverse leg is the `inverse` key.
# tools/configs.py holds the shapes and their algorithmic bytes.
def synth_u32(torch, batch, n, p, device, seed=SEED, first_row=0):
    """[batch][n] residues of a 32-bit modulus as int32 bit patterns: a[b][i] = splitmix64(seed + (first_row + b)*n + i) mod p.
    The unsigned 64-bit word u = hi*2^32 + lo is reduced as ((hi mod p) * (2^32 mod p) + lo) mod p, which fits int64 when
    (p - 1) * (2^32 mod p) + 2^32 < 2^63 (asserted)."""
    r32 = (1 << 32) % p
    assert p < (1 << 32) and (p - 1) * r32 + (1 << 32) < (1 << 63), "modulus outside the generator's int64 window"
    out = torch.empty((batch, n), dtype=torch.int32, device=device)
    rows = max(1, (1 << 24) // n)
    for r0 in range(0, batch, rows):
        r1 = min(batch, r0 + rows)
        z = splitmix64_words(torch, seed + (first_row + r0) * n, (r1 - r0) * n, device)
        hi, lo = (z >> 32) & 0xFFFFFFFF, z & 0xFFFFFFFF
        v = ((hi % p) * r32 + lo) % p
        out[r0:r1] = torch.where(v >= (1 << 31), v - (1 << 32), v).to(torch.int32).view(r1 - r0, n)
    return out


def rowsum_mod_p_u32(rows, p):
    """[sum(row) mod p] for uint32 rows (N <= 2^28 words of < 2^32 sum below 2^60)."""
    import numpy as np

    rows = np.ascontiguousarray(rows).view(np.uint32)
    return [int(v) % p for v in rows.sum(axis=1, dtype=np.uint64)]


def poly_eval_mod(coeffs, r, p):
    """sum(coeffs[i] * r^i) mod p by Horner's rule on Python integers (0.2-0.3 s per 2^20 coefficients)."""
    acc = 0
    for v in coeffs[::-1].tolist():
        acc = (acc * r + v) % p
    return acc


def events_ms(torch, fn, stream, steps, warmup):
    """median ms of `steps` calls of fn, one hipEvent pair per call on `stream`, and the ms per call of the same calls issued
    back to back inside one pair (what a resident pipeline of these operations takes)."""
    for _ in range(warmup):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for e0, e1 in evs:
        e0.record(stream)
        fn()
        e1.record(stream)
    b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b0.record(stream)
    for _ in range(steps):
        fn()
    b1.record(stream)
    stream.synchronize()
    each = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    return each[len(each) // 2], b0.elapsed_time(b1) / steps


def config_plain_share(c):
    """Share of plain moves / adds (2 cycles; the rest costs 4: tools/hw.py) among the VALU instructions of this configuration's
    butterfly statement, from the generator's instruction lists (tools/valu_mix.py; no GPU)."""
    from valu_mix import stream_mix

    if c["wb"] == 8:
        m = stream_mix("fwd" if c["p"] == GOLDILOCKS else "fwd64")
    else:
        m = stream_mix("fwd32:" + ("lazy" if c["p"] < (1 << 30) else "small" if c["p"] < (1 << 31) else "any"))
    return m["mix"].get("plain", 0) / float(m["valu"])


def config_roofline(cfg_key, c, op_ms, copy_ms, passes, src_hash):
    """bench.py's roofline keys for one BASELINE configuration (tools/configs.py): achieved = algorithmic bytes of one operation /
    time per operation; traffic and the vector-ALU fraction are quoted from profiles/rNN_<cfg>_pmc_traffic.json /
    _sq_counters.json (tools/collect_profiles.sh) only when their kernel-source hash equals this tree's."""
    from configs import algorithmic_bytes

    alg = float(algorithmic_bytes(c))
    plain = config_plain_share(c)
    # valu.frac_of_peak_at_held_clock: VALU wave-instructions per operation at their measured throughput (`plain` of them moves / adds) /
    # (1024 SIMDs x GRBM_GUI_ACTIVE / 8 per operation): clock-free
    achieved = alg / (op_ms * 1e-3) / 1e9
    pmc, pmc_src = tagged_profile("%s_pmc_traffic" % cfg_key, src_hash)
    sq, sq_src = tagged_profile("%s_sq_counters" % cfg_key, src_hash)
    traffic = pmc["per_op"]["hbm_bytes"] if pmc else None
    valu = None
    if cfg_key == "cfg5_shard" and (pmc is None or sq is None):
        # config 5's shard runs the headline's two kernels at twice the batch: quote THEIR counters (fractions are batch-free; bytes
        # per launch scale with the rows), and say so
        hp, hp_src = tagged_profile("pmc_traffic", src_hash)
        hs, hs_src = tagged_profile("sq_counters", src_hash)
        shape = [("contig", 0, 8), ("col", 8, 8)]
        if pmc is None and hp:
            ent, _ = forward_counters(hp, shape)
            if ent:
                traffic = sum(e[1]["hbm_bytes_per_launch"] for e in ent) * c["batch"] / float(hp.get("batch", 4096))
                pmc_src = hp_src + " -- the headline's launches of the same two kernels, scaled by the rows"
        if sq is None and hs:
            ent, _ = forward_counters(hs, shape)
            if ent and all(e[1].get("kernel_cycles") for e in ent):
                ins = sum(e[1]["SQ_INSTS_VALU"] for e in ent)
                cyc = sum(e[1]["kernel_cycles"] for e in ent)
                valu = {"instr_per_butterfly": ins / (hs["batch"] * (1 << (hs["logn"] - 1)) * hs["logn"] / 64.0),
                        "frac_of_peak_at_held_clock": valu_frac_of_peak(ins, cyc, plain),
                        "mean_waves_per_simd": [e[1].get("mean_waves_per_simd") for e in ent],
                        "held_clock_GHz": [e[1].get("held_clock_GHz") for e in ent], "kernels": [e[1]["short"] for e in ent],
                        "kernel_cycles": [e[1].get("kernel_cycles") for e in ent]}  # (the headline's two kernels: fractions do not depend on the batch)
                sq_src = hs_src + " -- the headline's launches of the same two kernels"
    if sq and sq["per_op"].get("kernel_cycles"):
        po = sq["per_op"]
        valu = {"instr_per_butterfly": po["valu_instr_per_butterfly"],
                "frac_of_peak_at_held_clock": valu_frac_of_peak(po["valu_instr"], po["kernel_cycles"], plain),
                "mean_waves_per_simd": [k.get("mean_waves_per_simd") for k in sq["kernels"].values()],
                "held_clock_GHz": [k.get("held_clock_GHz") for k in sq["kernels"].values()],
                "kernels": [k["short"] for k in sq["kernels"].values()],
                "kernel_cycles": [k.get("kernel_cycles") for k in sq["kernels"].values()]}
    if valu:
        # a clock above the part's peak is not a clock (sane_clocks): null it AND the fraction made of the same cycle count
        valu["held_clock_GHz"], note = sane_clocks(valu["held_clock_GHz"])
        if note:
            valu["held_clock_note"] = note
            valu["frac_of_peak_at_held_clock"] = None
    # physical trips through HBM on the convention the algorithmic bytes use: a transform moves 2N words per pass; the product's
    # fused schedule (inverse column passes of a and b 4N, fused middle 3N, forward column pass 2N) moves exactly the 9N it is priced on
    ceiling = 1.0 if c["op"] == "polymul" else 1.0 / max(1, passes)
    copy_gbs = alg / (copy_ms * 1e-3) / 1e9 if copy_ms else None
    of_copy = (achieved / ceiling / copy_gbs) if copy_gbs else None
    if op_ms < 0.05 and not (of_copy and of_copy >= 0.9):
        bound, why = "latency", ("one generation of workgroups: a %.1f us launch is a workgroup's own load -> butterflies -> store chain, "
                                 "not a throughput limit" % (op_ms * 1e3))
    else:
        clocks = [h for h in valu["held_clock_GHz"] if h] if valu else None
        wts = [c for h, c in zip(valu["held_clock_GHz"], valu.get("kernel_cycles") or []) if h] if valu else None
        bound, why = decide_bound([of_copy] if of_copy else None, valu["frac_of_peak_at_held_clock"] if valu else None,
                                  [w for w in valu["mean_waves_per_simd"] if w] if valu else None, clocks, wts)
    return {"bound": bound, "bound_detail": why, "roofline_of_fields": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "frac_ceiling": ceiling,
            "traffic": traffic, "traffic_ratio_to_algorithmic": (traffic / alg if traffic else None), "traffic_source": pmc_src,
            "valu": valu, "valu_source": sq_src,
            "algorithmic_bytes_per_op": alg, "device_copy_same_bytes_GBs": copy_gbs, "frac_of_device_copy": of_copy}


def run_config(torch, key, stream, src_hash, steps):
    """One BASELINE configuration end to end on the current device: plan, synthetic resident inputs (the SURVEY 8d generator),
    warm-up, `steps` timed operations, and an oracle-free verification of the result -- forward: inverse(forward(x)) == x over the
    whole batch and out[b][0] == sum(a[b][:]) mod p on sampled rows; product: c(r) == a(r) * b(r) mod p at a root r of x^N + 1 on
    sampled rows (Horner on Python integers: true for the negacyclic product and for nothing else) and, over the whole batch,
    InvU(c) == InvU(a) . InvU(b) word by word through the plain inverse passes and the pointwise kernel (other kernels than the
    fused product path ran)."""
    import numpy as np

    from configs import CONFIGS, algorithmic_bytes, butterflies
    from ntt_aie_amd import NTTPlan

    c = CONFIGS[key]
    logn, p, wb, batch = c["logn"], c["p"], c["wb"], c["batch"]
    n = 1 << logn
    dev = torch.device("cuda", torch.cuda.current_device())
    plan = NTTPlan(logn, p, wb, dev.index)
    plan.set_twiddles(plan.make_table(c["kind"], c["g"]))

    def synth(first_row):
        return synth_batch(torch, batch, n, dev, first_row=first_row) if wb == 8 else synth_u32(torch, batch, n, p, dev, first_row=first_row)

    # rows of the job's generator: the headline's batch never uses the first three ranges; config 5's shard is rank 3's rows of the
    # 8-GPU job ([3 * 8192, 4 * 8192) of its [65536][N] input)
    base_row = {"cfg2": 1 << 20, "cfg2_sat": 1 << 21, "cfg4": 1 << 22, "cfg5_shard": 3 * 8192}.get(key, 1 << 23)
    entry = {"name": c["name"], "key": key, "baseline_config": {"cfg2": 2, "cfg4": 4, "cfg5_shard": 5}.get(key), "logn": logn, "word_bytes": wb,
             "modulus": p, "batch": batch, "op": c["op"], "steps": steps, "data": "synthetic (splitmix64 mod p, rows %d.. of the job's generator)" % base_row}
    words = algorithmic_bytes(c) // 2 // wb
    if c["op"] == "forward":
        x = synth(base_row)
        y = torch.empty_like(x)
        med, b2b = events_ms(torch, lambda: plan.forward(x, y, stream=stream), stream, steps, 8)
        passes = plan.passes_for(batch)
        back = plan.inverse(y, stream=stream)
        stream.synchronize()
        rt = bool(torch.equal(back, x))
        rows = sorted(set(int(r) for r in np.linspace(0, batch - 1, min(16, batch))))
        idx = torch.tensor(rows, device=dev)
        xs = x.index_select(0, idx).cpu().numpy()
        y0 = y[:, 0].index_select(0, idx).cpu().numpy()
        if wb == 8:
            got, want = [int(v) for v in y0.view(np.uint64)], rowsum_mod_p(xs, p)
        else:
            got, want = [int(v) for v in y0.view(np.uint32)], rowsum_mod_p_u32(xs, p)
        sums = got == want and all(v < p for v in got)
        entry.update({"verified": bool(rt and sums),
                      "verification": {"round_trip_identical": rt, "coefficient_sum_invariant": bool(sums), "rows_sampled": len(rows)},
                      "hbm_passes": len(passes), "pass_stages": [st for _, _, st in passes]})
        del back
        npass = len(passes)
    else:
        a0, b0 = synth(base_row), synth(base_row + batch)
        a, b = a0.clone(), b0.clone()
        cbuf = torch.empty_like(a0)
        plan.polymul_negacyclic(a, b, cbuf, stream=stream)  # the verified product: fresh operands (the entry point overwrites them)
        stream.synchronize()
        r = pow(c["g"], (p - 1) // (2 * n), p)
        assert pow(r, n, p) == p - 1, "r must be a root of x^N + 1"
        rows = sorted(set(int(v) for v in np.linspace(0, batch - 1, 2)))
        ev = True
        for rr in rows:
            av, bv, cv = (t[rr].cpu().numpy().view(np.uint64) for t in (a0, b0, cbuf))
            ev = ev and bool(cv.max() < p) and poly_eval_mod(cv, r, p) == poly_eval_mod(av, r, p) * poly_eval_mod(bv, r, p) % p
        # whole batch: InvU(InvU(a) . InvU(b) . N^-1 -> Fwd) == InvU(a) . InvU(b); the unscaled inverse of a forward is N x identity
        ia = plan.inverse(a0, a, scale=False, stream=stream)   # a, b: scratch from here on
        ib = plan.inverse(b0, b, scale=False, stream=stream)
        prod = plan.pointwise_mul(ia, ib, ia, stream=stream)
        ic = plan.inverse(cbuf, ib, scale=False, stream=stream)
        stream.synchronize()
        dom = bool(torch.equal(prod, ic))
        # timed: operands in separate buffers, overwritten by every call (transform-domain words of the previous call: canonical residues)
        med, b2b = events_ms(torch, lambda: plan.polymul_negacyclic(a, b, cbuf, stream=stream), stream, steps, 2)
        entry.update({"verified": bool(ev and dom),
                      "verification": {"evaluation_at_root_of_xN_plus_1": ev, "rows_evaluated": len(rows),
                                       "transform_domain_identity_whole_batch": dom,
                                       "what": "c(r) == a(r) b(r) mod p, r = %d^((p-1)/2N) (Horner, Python integers) on %d rows; InvU(c) == "
                                               "InvU(a) . InvU(b) over all %d products through the plain passes + pointwise kernel" % (c["g"], len(rows), batch)},
                      "hbm_passes": plan.hbm_passes})
        npass = plan.hbm_passes
        del a0, b0, a, b, cbuf
    op_ms = min(med, b2b)
    src = torch.empty(words, dtype=torch.int64 if wb == 8 else torch.int32, device=dev)
    dst = torch.empty_like(src)
    copy_ms = device_copy_rate(torch, src, dst, stream)["ms"]  # a copy of the operation's algorithmic bytes, best of both forms
    del src, dst
    unit = "products/s" if c["op"] == "polymul" else "NTT/s"
    entry.update({"ms": op_ms, "ms_median_single": med, "ms_back_to_back": b2b, "value": batch / (op_ms * 1e-3), "unit": unit,
                  "butterflies_per_s": butterflies(c) / (op_ms * 1e-3),
                  "roofline": config_roofline(key, c, op_ms, copy_ms, npass, src_hash)})
    plan.close()
    return entry


def configs_verdict(entries):
    """(every entry verified?, exit code).  An entry whose result was compared and found WRONG makes the exit code 1 (no time
    without a check); an entry that could not run at all (an exception: reported with its error, counted as not verified) does
    not -- the headline's own numbers stand and the line says which leg failed.  Pure (CPU unit test)."""
    all_ok = all(bool(e.get("verified")) for e in entries)
    wrong = any((not e.get("verified")) and "error" not in e for e in entries)
    return all_ok, (1 if wrong else 0)


def extra_configs(torch, stream, src_hash, steps=10):
    """[entry per config] and whether every entry that ran verified; an entry that could not run (exception) is reported with
    its error and counts as NOT verified -- it never stops the headline line from being printed."""
    out = []
    for key in ("cfg2", "cfg4", "cfg5_shard"):
        try:
            with torch.cuda.stream(stream):
                out.append(run_config(torch, key, stream, src_hash, steps))
        except Exception as e:  # reported in the line; the headline's own numbers stand
            sys.stderr.write("bench.py: config %s failed: %r\n" % (key, e))
            out.append({"key": key, "verified": False, "error": repr(e)})
        torch.cuda.empty_cache()
    return out


def run_single_process(args):
    """One process, N devices at the C boundary (the reference scatters, broadcasts its table and gathers below
    ONE host, src/aie2.py:83-115): one plan on device 0, ntt_plan_clone onto devices 1..N-1 (tables device-to-device), one
    stream and one [batch][N] shard per device, an event pair per device.  Aggregate = total NTT / the timed region, which
    ends when the slowest device does.  NTT_BENCH_ONE_DEVICE=1 places every clone on device 0 (rehearsal on a one-GPU box)."""
    import torch

    from ntt_aie_amd.multi import MultiDevicePlan

    ndev = args.gpus
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    rehearsal = os.environ.get("NTT_BENCH_ONE_DEVICE") == "1"
    have = torch.cuda.device_count()
    if not rehearsal and have < ndev:
        sys.stderr.write("bench.py: --gpus %d --single-process but only %d device(s) visible (NTT_BENCH_ONE_DEVICE=1 rehearses "
                         "the path with every clone on device 0)\n" % (ndev, have))
        return 2
    devs = [0] * ndev if rehearsal else list(range(ndev))
    logn, n, batch, p = args.logn, 1 << args.logn, args.batch, GOLDILOCKS
    mdp = MultiDevicePlan(logn, p, 8, devs)  # one plan on the first device ...
    table = mdp.make_table(0, 7)
    mdp.set_twiddles(table)                  # ... ntt_plan_clone onto the others: hipMemcpyPeer of the tables when the device differs
    plans, streams, plan0 = mdp.plans, mdp.streams, mdp.plans[0]
    xs, ys = [], []
    for i, d in enumerate(devs):
        with torch.cuda.device(d):
            xs.append(synth_batch(torch, batch, n, torch.device("cuda", d), first_row=i * batch))  # shard i = rows [i*batch, (i+1)*batch) of the job
            ys.append(torch.empty_like(xs[-1]))
            torch.cuda.synchronize(d)  # the generator ran on the device's default stream; the transforms run on the shard's own stream

    def step():
        for pl, x, y, st in zip(plans, xs, ys, streams):
            pl.forward(x, y, stream=st)

    def sync_all():
        for st in streams:
            st.synchronize()

    for _ in range(PREWARM + args.warmup):
        step()
    sync_all()
    ev = []
    for d, st in zip(devs, streams):
        with torch.cuda.device(d):
            ev.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
    t0 = time.perf_counter()
    for (e0, _), st, d in zip(ev, streams, devs):
        with torch.cuda.device(d):
            e0.record(st)
    for _ in range(args.steps):
        step()
    for (_, e1), st, d in zip(ev, streams, devs):
        with torch.cuda.device(d):
            e1.record(st)
    sync_all()
    elapsed = time.perf_counter() - t0
    dev_ms = [e0.elapsed_time(e1) / args.steps for e0, e1 in ev]
    value = batch * ndev * args.steps / elapsed
    passes = plan0.passes_for(batch)
    out = base_line(args, logn, batch, ndev, value, elapsed, passes,
                    "ntt_plan_clone: tables copied device-to-device (hipMemcpyPeer)" if ndev > 1 else "none (one device)",
                    "single-process")
    out["value_by_device_events"] = batch * ndev / (max(dev_ms) * 1e-3)
    # every device proves its shard, exactly as every rank does in the process-per-GPU form
    recs, flags = [], [True, True]
    for i, (d, pl, x, y, st) in enumerate(zip(devs, plans, xs, ys, streams)):
        with torch.cuda.device(d):
            v = verify_shard(torch, pl, x, y, p, st)
        flags = [flags[0] and v["round_trip_identical"], flags[1] and v["coefficient_sum_invariant"]]
        recs.append(dict(device_identity(torch, d), rank=i, ms_per_step=dev_ms[i], **v))
    fields, code = verdict_fields(flags, recs, ndev, one_device_ok=rehearsal)
    out.update(fields)
    with torch.cuda.device(devs[0]), torch.cuda.stream(streams[0]):  # torch's own kernels (the device copy) on the launch stream too
        code = max(code, rank0_extras(torch, args, plan0, table, xs[0], ys[0], streams[0], passes, out, ndev))
    emit(out, args)
    return code


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--logn", type=int, default=16)
    ap.add_argument("--batch", type=int, default=4096, help="polynomials per GPU")
    ap.add_argument("--single-process", action="store_true",
                    help="one process drives all --gpus devices (ntt_plan_clone, one stream per device) instead of one rank per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline's multi-core leg (default: every core of the affinity mask)")
    ap.add_argument("--no-valu-floor", action="store_true",
                    help="skip the VALU-floor leg (its launches carry the same kernel names: keep them out of a rocprofv3 --stats run)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip BASELINE configs 2 and 4 (counter collections: the headline's launches alone in the profile)")
    ap.add_argument("--no-inverse", action="store_true",
                    help="skip the inverse-transform leg (counter collection: only forward kernels in the profile)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) even for one rank: exercises the twiddle broadcast path")
    ap.add_argument("--explain", action="store_true",
                    help="keep the explanatory strings (what / definition / bound_note / full provenance) and full float precision in the "
                         "line; the default line is numbers only, < 6 KB (DESIGN.md section 4 says what every key means)")
    ap.add_argument("--rdzv-file", default=None,
                    help="rendezvous through this file (torch.distributed FileStore) instead of MASTER_ADDR / MASTER_PORT: no TCP "
                         "port to agree on; what `python bench.py --gpus N` uses for the ranks it starts itself")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.single_process:
        if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
            raise SystemExit("--single-process under a multi-rank launcher: start it as plain `python bench.py --gpus N --single-process`")
        sys.exit(run_single_process(args))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args, sys.argv[1:]))

    # the host driver only supports dmabuf IPC: RCCL across processes needs this (already exported on the GPU boxes)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start it as `python bench.py --gpus N` (self-launching) or under "
                         "torch.distributed.run with --nproc-per-node N" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # rehearsal on a one-GPU box (tests only): NTT_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and
    # NTT_BENCH_BACKEND=gloo replaces RCCL, which refuses two ranks on one device
    if os.environ.get("NTT_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    elif world > 1 and torch.cuda.device_count() == 1:
        # a launcher that gives every rank its OWN visible device (ROCR_/HIP_VISIBLE_DEVICES per rank): device 0 of this process.
        # Ranks that in fact share one GPU are caught below: the line is verified only on `world` distinct PCI ids.
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d: local rank %d but %d device(s) visible" % (rank, local_rank, torch.cuda.device_count()))
    backend = os.environ.get("NTT_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        # a launcher (torchrun, the driver) supplies MASTER_ADDR / MASTER_PORT; without one the rendezvous is a file
        # (--rdzv-file; a one-rank --force-dist run without either makes its own temporary one): no port is ever guessed
        kw = {}
        if args.rdzv_file or "MASTER_PORT" not in os.environ:
            import tempfile

            path = args.rdzv_file or os.path.join(tempfile.mkdtemp(prefix="ntt_rdzv_"), "store")
            if world > 1 and not args.rdzv_file:
                raise SystemExit("multi-rank run without MASTER_ADDR/MASTER_PORT or --rdzv-file: nothing to rendezvous on")
            kw["init_method"] = "file://" + path
        else:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, **kw)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)

    from ntt_aie_amd.dist import ShardedNTT

    logn, n, batch, p = args.logn, 1 << args.logn, args.batch, GOLDILOCKS
    eng = ShardedNTT(logn, p, g=7, word_bytes=8, device=local_rank)  # rank 0 makes the table, RCCL broadcast
    plan = eng.plan
    # rank r holds rows [r*batch, (r+1)*batch) of the job's [world*batch][N] input
    x = synth_batch(torch, batch, n, dev, first_row=rank * batch)
    y = torch.empty_like(x)
    stream = torch.cuda.current_stream()

    for _ in range(PREWARM + args.warmup):
        plan.forward(x, y, stream=stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        plan.forward(x, y, stream=stream)
    torch.cuda.synchronize()
    own = time.perf_counter() - t0  # this rank's own K steps, before it waits for the others
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # no time without a check: every rank proves its own shard and says which device it ran on
    v = verify_shard(torch, plan, x, y, p, stream)
    # test hook, honoured in the one-device rehearsal only (NTT_BENCH_ONE_DEVICE=1): this rank REPORTS a failed check, so that the
    # tests can watch one bad rank turn the line red; it can never turn a failure into a pass
    if os.environ.get("NTT_BENCH_ONE_DEVICE") == "1" and os.environ.get("NTT_BENCH_INJECT_FAILURE") == str(rank):
        v["round_trip_identical"] = False
    ident = dict(device_identity(torch, local_rank), rank=rank, ms_per_step=own / args.steps * 1e3, **v)
    reduced, recs = reduce_verdicts(dist, torch, dev, world, rank, [v["round_trip_identical"], v["coefficient_sum_invariant"]], ident)
    fields, code = verdict_fields(reduced, recs, dist.get_world_size() if use_dist else 1,
                                  one_device_ok=os.environ.get("NTT_BENCH_ONE_DEVICE") == "1")

    total_ntt = batch * world * args.steps
    value = total_ntt / elapsed
    passes = plan.passes_for(batch)  # the decomposition the launcher picks for THIS batch (plan alternatives)
    out = base_line(args, logn, batch, world, value, elapsed, passes,
                    (dist.get_backend() if use_dist else "none (single process)"), "process-per-gpu")
    out.update(fields)

    if rank == 0:
        code = max(code, rank0_extras(torch, args, plan, eng.table, x, y, stream, passes, out, world))
        emit(out, args)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(code)


if __name__ == "__main__":
    main()
