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
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s achievable by a float4 copy
SEED = 0x9E3779B97F4A7C15  # a[b][i] = splitmix64(SEED + b*N + i) mod p
PROFILE_ROUND = "r04"
SIMDS = 1024            # 256 CUs x 4 SIMDs
PEAK_CLOCK_GHZ = 2.4    # MI355X_MICROARCH.md: peak engine clock
VALU_CYCLES_PER_WAVE_INSTR = 4  # one wave64 instruction on a 16-lane SIMD (assumed for EVERY VALU form: see roofline.valu.what)


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


def cpu_baseline(logn, p, table, rows_fn, cpu_seconds=20.0, threads=None):
    """Oracle (port of the reference CPU verification path, src/test.cpp:34-60: three `%` per butterfly) on the host cores,
    bounded sample: one thread (the reference is single-threaded) and ALL the cores this process may run on,
    one polynomial per task; the count is stated next to the figure.

    The sample is the GPU's own input (the reference runs its CPU path on the same a[i] it handed the device,
    src/test.cpp:203-207): `rows_fn(k)` returns rows [0, k) of the job's synthetic batch as a host uint64 array -- the first
    rows of the resident device buffer copied back, continued by the same generator when the sample is longer than the batch."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O

    n = 1 << logn
    aff, quota = host_cores()
    # every core this process can actually run on: the affinity mask, capped by the cgroup CPU quota (on the GPU boxes the mask
    # shows all 256 host threads but the container is throttled to 16 cores: 256 OpenMP threads then run slower than 16)
    avail = aff if not quota else max(1, min(aff, int(quota + 0.999)))
    cores = int(threads) if threads else avail
    probe = rows_fn(2)
    t0 = time.perf_counter()
    O.ntt(probe, table, p, nthreads=1)
    t1 = (time.perf_counter() - t0) / 2
    rate_1 = 1.0 / t1
    # bounded sample: about `cpu_seconds` of CPU work in total, spread over the host threads (at least 4 polynomials per thread)
    sample = int(max(cores * 4, min(16384, cpu_seconds / t1)))
    a = rows_fn(sample)
    t0 = time.perf_counter()
    O.ntt(a, table, p, nthreads=cores)
    tn = time.perf_counter() - t0
    rate_n = sample / tn
    best, used = (rate_n, cores) if rate_n >= rate_1 else (rate_1, 1)
    return {"value": best, "unit": "NTT/s", "cores": used, "kind": "port",
            "sample": "rows 0..%d of the GPU's own input batch (same seed, same generator, same table), N=2^%d, on %d threads "
                      "(%.2f s) -- every core available to this process: affinity mask %d%s; 1-thread rate %.1f NTT/s"
                      % (sample - 1, logn, cores, tn, aff,
                         (", cgroup CPU quota %.1f cores" % quota) if quota else ", no cgroup quota", rate_1),
            "sample_rows": sample, "sample_is_gpu_input": True,
            "host_affinity_cores": aff, "host_cgroup_quota_cores": quota, "host_available_cores": avail, "threads_all_cores_leg": cores,
            "value_all_cores": rate_n, "value_1thread": rate_1, "butterflies_per_s": best * (n // 2) * logn}


def device_copy_rate(torch, x, y, stream, reps=10):
    """The achievable stream rate of this device, in this process, on these buffers: read every byte of x once and
    write it once to y (the same algorithmic bytes as one transform).  Two forms, best kept: the runtime's
    device-to-device copy and a plain elementwise kernel."""
    def timed(fn):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

    nbytes = 2.0 * x.numel() * x.element_size()
    ms_copy = timed(lambda: y.copy_(x))
    ms_elem = timed(lambda: torch.bitwise_xor(x, 1, out=y))
    best = min(ms_copy, ms_elem)
    return {"GBs": nbytes / (best * 1e-3) / 1e9, "ms": best, "ms_memcpy_d2d": ms_copy, "ms_elementwise_kernel": ms_elem,
            "bytes": nbytes}


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
    """profiles/<round>_<name>.json if it was collected on exactly these kernel sources, else (None, reason)."""
    path = os.path.join(ROOT, "profiles", "%s_%s.json" % (PROFILE_ROUND, name))
    if not os.path.exists(path):
        return None, "profiles/%s_%s.json absent" % (PROFILE_ROUND, name)
    d = json.load(open(path))
    if d.get("src_hash") != src_hash:
        return None, "profiles/%s_%s.json was collected on kernel sources %s, this tree is %s: not quoted" % (
            PROFILE_ROUND, name, d.get("src_hash"), src_hash)
    return d, "profiles/%s_%s.json (src_hash %s)" % (PROFILE_ROUND, name, src_hash)


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


def weighted_issue_cycles(stream_mix, instr_per_butterfly, costs, overhead_cycles=4.0, default_cycles=4.0):
    """Issue cycles per wave-butterfly of one pass kernel: the butterfly stream's instructions priced by class with MEASURED
    cycles per wave-instruction (profiles/rNN_valu_issue_cost.json), plus the instructions the counter sees beyond the stream
    (SQ_INSTS_VALU per butterfly - stream length: addressing, register moves) at `overhead_cycles` each.
    stream_mix = {"valu": n, "mix": {class: count}}; costs = {class: cycles}.  Pure arithmetic (CPU unit test)."""
    stream = sum(n * costs.get(c, default_cycles) for c, n in stream_mix["mix"].items())
    extra = max(0.0, instr_per_butterfly - stream_mix["valu"])
    return stream + extra * overhead_cycles


def valu_roofline(sq_entries, passes, per_pass_ms, batch, logn, issue_model=None):
    """The vector-ALU roofline of the forward transform (the unit that binds).
    peak butterflies/s = SIMDs x clock / (4 cycles x VALU instructions per butterfly) x 64 lanes.
    issue_model = {"costs": {class: cycles}, "streams": [stream mix per pass], "source": "..."} adds the WEIGHTED figure:
    every instruction class priced at its measured issue cost instead of a flat 4 cycles."""
    n = 1 << logn
    bf = [batch * (n // 2) * stages for _, _, stages in passes]            # butterflies per launch of each pass
    ipb = [e[1]["valu_instr_per_butterfly"] for e in sq_entries]          # SQ_INSTS_VALU / wave-butterflies, forward kernels
    mean_ipb = sum(i * b for i, b in zip(ipb, bf)) / sum(bf)
    held = [e[1].get("held_clock_GHz") for e in sq_entries]
    cyc = [e[1].get("kernel_cycles") for e in sq_entries]
    achieved = sum(bf) / (sum(per_pass_ms) * 1e-3)

    def peak(f_ghz, instr):
        return SIMDS * f_ghz * 1e9 / (VALU_CYCLES_PER_WAVE_INSTR * instr) * 64

    out = {
        "instr_per_butterfly": ipb, "instr_per_butterfly_mean": mean_ipb,
        "peak_butterflies_per_s": peak(PEAK_CLOCK_GHZ, mean_ipb), "peak_clock_GHz": PEAK_CLOCK_GHZ,
        "achieved_butterflies_per_s": achieved,
        "frac_at_2.4GHz": achieved / peak(PEAK_CLOCK_GHZ, mean_ipb),
        "frac_at_2.4GHz_per_pass": [b / (t * 1e-3) / peak(PEAK_CLOCK_GHZ, i) for b, t, i in zip(bf, per_pass_ms, ipb)],
        "kernels": [e[0] for e in sq_entries],
        "what": "instr_per_butterfly = SQ_INSTS_VALU of the FORWARD pass kernels / (butterflies / 64); peak = %d SIMDs x f / "
                "(%d cycles x instr) x 64 lanes, every VALU form priced at %d cycles; frac_at_2.4GHz uses THIS run's pass durations; "
                "frac_at_held_clock is clock-free: instr x 4 cycles x wave-butterflies / (SIMDs x GRBM_GUI_ACTIVE/8) of the "
                "counter run, with held_clock_GHz = GRBM_GUI_ACTIVE / 8 / duration of the same profiled launches; "
                "frac_at_held_clock_weighted replaces the flat 4 cycles by the measured issue cost of each instruction class "
                "(tools/valu_issue_cost.hip: plain ops retire in ~2 cycles, VOP3 carry forms in ~4, v_mad_u64_u32 in more)"
                % (SIMDS, VALU_CYCLES_PER_WAVE_INSTR, VALU_CYCLES_PER_WAVE_INSTR),
    }
    if all(h for h in held) and all(c for c in cyc):
        out["held_clock_GHz"] = held
        out["frac_at_held_clock_per_pass"] = [e[1]["valu_instr_x4cyc_over_kernel_cycles"] for e in sq_entries]
        tot_cyc = sum(cyc)
        out["frac_at_held_clock"] = sum(i * b / 64 for i, b in zip(ipb, bf)) * VALU_CYCLES_PER_WAVE_INSTR / (SIMDS * tot_cyc)
        # the clock THIS run held, if a launch takes the same number of cycles as under the profiler
        out["clock_this_run_GHz_estimate"] = [c / (t * 1e6) for c, t in zip(cyc, per_pass_ms)]
        if issue_model:
            wcyc = [weighted_issue_cycles(m, i, issue_model["costs"], issue_model.get("overhead_cycles", 4.0))
                    for m, i in zip(issue_model["streams"], ipb)]
            per_pass = [w * b / 64 / (SIMDS * c) for w, b, c in zip(wcyc, bf, cyc)]
            out["issue_cycles_per_butterfly_weighted"] = wcyc
            out["frac_at_held_clock_weighted_per_pass"] = per_pass
            out["frac_at_held_clock_weighted"] = sum(w * b / 64 for w, b in zip(wcyc, bf)) / (SIMDS * tot_cyc)
            out["issue_model"] = {"class_cycles": issue_model["costs"], "stream_mix": [m["mix"] for m in issue_model["streams"]],
                                  "overhead_cycles": issue_model.get("overhead_cycles", 4.0), "source": issue_model.get("source"),
                                  "measured_stream_cycles_per_butterfly": issue_model.get("measured_stream_cycles")}
            out["saturated"] = bool(out["frac_at_held_clock_weighted"] >= 0.97)
            out["verdict"] = ("vector-ALU issue capacity is %.0f %% used at the held clock: saturated, an instruction saved returns as time"
                              % (100 * out["frac_at_held_clock_weighted"]) if out["saturated"] else
                              "vector-ALU issue capacity is %.0f %% used at the held clock: the rest is issue stalls (SQ_WAIT_INST_ANY %s of "
                              "wave-cycles in the counter run: dependent carry chains and LDS / memory waits with 3.7 waves per SIMD)"
                              % (100 * out["frac_at_held_clock_weighted"],
                                 "/".join("%.2f" % e[1].get("wave_issue_stall_frac", float("nan")) for e in sq_entries)))
    return out


def load_issue_model(passes, src_hash, waves_per_simd=4):
    """The weighted VALU model's inputs, or (None, reason): profiles/<round>_valu_issue_cost.json (measured on the GPU box by
    tools/valu_issue_cost) and profiles/<round>_valu_mix.json (tools/valu_mix.py, stamped with the kernel-source hash)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    cost_path = os.path.join(ROOT, "profiles", "%s_valu_issue_cost.json" % PROFILE_ROUND)
    if not os.path.exists(cost_path):
        return None, "profiles/%s_valu_issue_cost.json absent" % PROFILE_ROUND
    mix, why = tagged_profile("valu_mix", src_hash)
    if mix is None:
        return None, why
    from valu_mix import class_costs

    ic = json.load(open(cost_path))
    costs = class_costs(ic, waves_per_simd)
    # first pass: per-lane twiddles (VGPRs); column passes: wave-uniform twiddles (SGPRs)
    streams = [mix["streams"]["gl_fwd_v" if kind == "contig" else "gl_fwd_s"] for kind, _, _ in passes]
    meas = {k: v["cycles_per_butterfly"].get(str(waves_per_simd)) for k, v in ic.get("streams", {}).items()}
    return {"costs": costs, "streams": streams, "overhead_cycles": costs.get("other", 4.0), "measured_stream_cycles": meas,
            "source": "profiles/%s_valu_issue_cost.json at %d waves per SIMD + %s" % (PROFILE_ROUND, waves_per_simd, why)}, None


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


def verdict_fields(reduced, recs, world_seen):
    """The line's verification block and the process exit code (non-zero when ANY rank failed either check)."""
    ok = all(reduced)
    ids = [r.get("pci_bus_id") or r.get("uuid") for r in recs]
    return {"all_ranks_verified": ok, "world_size_seen": int(world_seen),
            "verification": {"round_trip_identical_all": reduced[0], "coefficient_sum_invariant_all": reduced[1],
                             "what": "every rank: inverse(forward(x)) == x over its whole shard (torch.equal) and out[b][0] == "
                                     "sum(a[b][:]) mod p on sampled rows; flags reduced with all_reduce(MIN)"},
            "distinct_devices": len(set(i for i in ids if i)) if any(ids) else None,
            "ranks": recs}, (0 if ok else 1)


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: the parent -- before anything touches the GPU -- starts the N ranks
    as `python -m torch.distributed.run ... bench.py <same arguments>` in a fresh child process, relays the child's
    output (rank 0's ONE JSON line) and exit code.  The torchrun form keeps working: with WORLD_SIZE set this is never reached."""
    import socket
    import subprocess

    rehearsal = os.environ.get("NTT_BENCH_ONE_DEVICE") == "1"
    if not rehearsal:
        import torch  # device_count() does not initialise the GPU on this image

        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.stderr.write("bench.py: --gpus %d but only %d device(s) visible; refusing to oversubscribe a GPU "
                             "(NTT_BENCH_ONE_DEVICE=1 NTT_BENCH_BACKEND=gloo rehearses the launch path on one device)\n"
                             % (args.gpus, have))
            return 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = None
    for _ in range(4):
        # a free port is probed, released and handed to the launcher: another process can take it in between (EADDRINUSE) --
        # then, and only then, the launch is repeated on a new port
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
        proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        collided = proc.returncode != 0 and ("EADDRINUSE" in proc.stderr or "address already in use" in proc.stderr.lower())
        if not collided:
            break
    sys.stderr.write(proc.stderr)
    sys.stdout.write(proc.stdout)
    sys.stdout.flush()
    return proc.returncode


def config_name(logn, batch, world):
    if logn == 16 and batch == 4096:
        return "BASELINE config 3's forward leg = the headline metric (N=2^16 Goldilocks, batch 4096 on one MI355X)" + (
            ", weak-scaled: 4096 per GPU" if world > 1 else "")
    if logn == 16 and batch == 8192:
        return ("BASELINE config 5 (N=2^16 Goldilocks, batch 65536 sharded across 8 MI355X = 8192 per GPU): %d GPU(s) x 8192 = %d "
                "polynomials in this job" % (world, world * batch))
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
        "config": {"workload": "%s; N=2^%d forward NTT, p=2^64-2^32+1, make_roots table g=7, batch=%d per GPU (%d in the job), "
                               "out-of-place, inputs resident in HBM, a[b][i] = splitmix64(0x9E3779B97F4A7C15 + b*N + i) mod p"
                               % (config_name(logn, batch, world), logn, batch, batch * world),
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
    (N = 1 only) and config 3's inverse leg.  Overwrites y."""
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
                model, model_why = load_issue_model(passes, src_hash)
                valu = valu_roofline(ent, passes, [float(v) for v in per_pass], batch, logn, issue_model=model)
                if model is None:
                    valu["issue_model"] = None
                    valu["issue_model_source"] = "weighted figure not computed: " + model_why
            else:
                valu_src += "; not quoted: " + why
    step_s = out["ms_per_step"] * 1e-3
    out["roofline"] = {
        # The contract's figure: algorithmic HBM bytes / kernel time against the 8 TB/s spec peak (achieved, peak, unit, frac).
        # `bound` names the unit that actually binds this integer kernel: the vector ALU, whose own roofline
        # is the `valu` object; `frac_ceiling` is what `frac` could reach at most with this pass count.
        "bound": "valu",
        "bound_note": "achieved/peak/frac are the HBM roofline SURVEY 8(d) prescribes (algorithmic bytes over the spec peak); the "
                      "binding unit is the vector ALU under the 1400 W board power cap (roofline.valu: instructions per "
                      "butterfly against 1024 SIMDs x clock / measured issue cycles; profiles/%s_power_probe.txt for the clock the cap "
                      "allows) -- each pass streams at ~0.87 of the device-copy rate, so HBM is the second constraint, not the first"
                      % PROFILE_ROUND,
        "frac_ceiling": 1.0 / len(passes),
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        # the same bytes over the step time the line's own `value` is made of (launch gaps included).  A step cannot be shorter
        # than its kernels, so frac_step is capped at frac; the two are measured seconds apart, and when the uncapped quotient
        # (frac_step_uncapped) comes out above frac the difference is the clock the chip held in each phase, not a faster step
        "frac_step": min(alg_bytes / step_s / 1e9, achieved) / HBM_PEAK_GBS,
        "frac_step_uncapped": alg_bytes / step_s / 1e9 / HBM_PEAK_GBS,
        "achieved_step": alg_bytes / step_s / 1e9,
        "traffic": traffic, "traffic_source": traffic_src,
        "definition": "algorithmic bytes of one forward transform (2*N*8 B) x batch / summed duration of its "
                      "%d pass kernels (hipEvents on the launch stream); frac_step divides by ms_per_step of the timed region instead "
                      "(kernel gaps included; on a multi-rank job the slowest rank's); a %d-pass transform physically moves %dx its "
                      "algorithmic bytes, so its ceiling is frac %.2f; traffic = PMC HBM bytes of the same (forward) launches"
                      % (len(passes), len(passes), len(passes), 1.0 / len(passes)),
        "algorithmic_bytes_per_transform": 2 * n * 8, "algorithmic_bytes_per_launch": alg_bytes,
        "passes": len(passes), "pass_stages": [stages for _, _, stages in passes],
        "pass_ms": [float(v) for v in per_pass], "dominant_pass": dom,
        # each pass kernel reads and writes every coefficient once: its own stream rate
        "pass_stream_GBs": [alg_bytes / (float(v) * 1e-3) / 1e9 for v in per_pass],
        "pass_stream_frac": [alg_bytes / (float(v) * 1e-3) / 1e9 / HBM_PEAK_GBS for v in per_pass],
        # the same bytes through a plain copy, same process, same buffers: the achievable rate beside the spec peak
        "device_copy": copy, "frac_of_device_copy": achieved / copy["GBs"],
        "pass_stream_frac_of_device_copy": [alg_bytes / (float(v) * 1e-3) / 1e9 / copy["GBs"] for v in per_pass],
        # the binding unit of this integer workload is the vector ALU: floor = the same kernels, loads from L2, no stores
        "valu_floor_pass_ms": floor,
        "valu_floor_frac_of_pass": ([f / float(v) for f, v in zip(floor, per_pass)] if floor else None),
        "valu_floor_source": floor_src,
        "valu": valu, "valu_source": valu_src,
    }
    if world == 1 and not args.no_cpu_baseline:
        def rows_fn(k):
            # rows [0, k) of the job's input: the head of the resident buffer, continued by the same generator beyond the batch
            head = x[:min(k, batch)].cpu().numpy().view(np.uint64)
            if k <= batch:
                return np.ascontiguousarray(head)
            more = synth_batch(torch, k - batch, n, x.device, first_row=batch).cpu().numpy().view(np.uint64)
            return np.concatenate([head, more])

        out["cpu_baseline"] = cpu_baseline(logn, p, table, rows_fn, threads=args.cpu_threads or None)
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
    fields, code = verdict_fields(flags, recs, ndev)
    out.update(fields)
    with torch.cuda.device(devs[0]), torch.cuda.stream(streams[0]):  # torch's own kernels (the device copy) on the launch stream too
        rank0_extras(torch, args, plan0, table, xs[0], ys[0], streams[0], passes, out, ndev)
    print(json.dumps(out), flush=True)
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
    ap.add_argument("--no-inverse", action="store_true",
                    help="skip the inverse-transform leg (counter collection: only forward kernels in the profile)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) even for one rank: exercises the twiddle broadcast path")
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
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d: local rank %d but %d device(s) visible" % (rank, local_rank, torch.cuda.device_count()))
    backend = os.environ.get("NTT_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

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
    fields, code = verdict_fields(reduced, recs, dist.get_world_size() if use_dist else 1)

    total_ntt = batch * world * args.steps
    value = total_ntt / elapsed
    passes = plan.passes_for(batch)  # the decomposition the launcher picks for THIS batch (plan alternatives)
    out = base_line(args, logn, batch, world, value, elapsed, passes,
                    (dist.get_backend() if use_dist else "none (single process)"), "process-per-gpu")
    out.update(fields)

    if rank == 0:
        rank0_extras(torch, args, plan, eng.table, x, y, stream, passes, out, world)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(code)


if __name__ == "__main__":
    main()
