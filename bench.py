#!/usr/bin/env python3
"""bench.py -- forward-NTT throughput of the MI355X engine on BASELINE.json's headline
configuration: N = 2^16, Goldilocks prime 2^64-2^32+1, batch 4096 per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one forward transform of the whole resident batch (input buffer -> output
buffer, both already in HBM).  Weak scaling: every rank owns `batch` polynomials and never
talks to the others in the timed region; the only collective is the one-off twiddle-table
broadcast from rank 0 (RCCL).  Rank 0 prints ONE JSON line.

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
PROFILE_ROUND = "r03"
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


def cpu_baseline(logn, p, table, cpu_seconds=20.0, threads=None):
    """Oracle (port of the reference CPU verification path, src/test.cpp:34-60: three `%` per butterfly) on the host cores,
    bounded sample: one thread (the reference is single-threaded) and ALL the cores this process may run on,
    one polynomial per task; the count is stated next to the figure."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle_py as O

    n = 1 << logn
    aff, quota = host_cores()
    # every core this process can actually run on: the affinity mask, capped by the cgroup CPU quota (on the GPU boxes the mask
    # shows all 256 host threads but the container is throttled to 16 cores: 256 OpenMP threads then run slower than 16)
    avail = aff if not quota else max(1, min(aff, int(quota + 0.999)))
    cores = int(threads) if threads else avail
    rng = np.random.default_rng(1)
    probe = rng.integers(0, 2**63, size=(2, n), dtype=np.uint64)
    t0 = time.perf_counter()
    O.ntt(probe, table, p, nthreads=1)
    t1 = (time.perf_counter() - t0) / 2
    rate_1 = 1.0 / t1
    # bounded sample: about `cpu_seconds` of CPU work in total, spread over the host threads (at least 4 polynomials per thread)
    sample = int(max(cores * 4, min(16384, cpu_seconds / t1)))
    a = rng.integers(0, 2**63, size=(sample, n), dtype=np.uint64)
    t0 = time.perf_counter()
    O.ntt(a, table, p, nthreads=cores)
    tn = time.perf_counter() - t0
    rate_n = sample / tn
    best, used = (rate_n, cores) if rate_n >= rate_1 else (rate_1, 1)
    return {"value": best, "unit": "NTT/s", "cores": used, "kind": "port",
            "sample": "%d polynomials of N=2^%d on %d threads (%.2f s) -- every core available to this process: affinity mask %d%s; "
                      "1-thread rate %.1f NTT/s" % (sample, logn, cores, tn, aff,
                                                    (", cgroup CPU quota %.1f cores" % quota) if quota else ", no cgroup quota", rate_1),
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


def valu_roofline(sq_entries, passes, per_pass_ms, batch, logn):
    """The vector-ALU roofline of the forward transform (the unit that binds).
    peak butterflies/s = SIMDs x clock / (4 cycles x VALU instructions per butterfly) x 64 lanes."""
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
                "(%d cycles x instr) x 64 lanes, every VALU form priced at %d cycles (v_mad_u64_u32 issues slower, plain "
                "add/mov faster: profiles/r01_microbench_valu_rates.txt); frac_at_2.4GHz uses THIS run's pass durations; "
                "frac_at_held_clock is clock-free: instr x 4 cycles x wave-butterflies / (SIMDs x GRBM_GUI_ACTIVE/8) of the "
                "counter run, with held_clock_GHz = GRBM_GUI_ACTIVE / 8 / duration of the same profiled launches"
                % (SIMDS, VALU_CYCLES_PER_WAVE_INSTR, VALU_CYCLES_PER_WAVE_INSTR),
    }
    if all(h for h in held) and all(c for c in cyc):
        out["held_clock_GHz"] = held
        out["frac_at_held_clock_per_pass"] = [e[1]["valu_instr_x4cyc_over_kernel_cycles"] for e in sq_entries]
        tot_cyc = sum(cyc)
        out["frac_at_held_clock"] = sum(i * b / 64 for i, b in zip(ipb, bf)) * VALU_CYCLES_PER_WAVE_INSTR / (SIMDS * tot_cyc)
        # the clock THIS run held, if a launch takes the same number of cycles as under the profiler
        out["clock_this_run_GHz_estimate"] = [c / (t * 1e6) for c, t in zip(cyc, per_pass_ms)]
    return out


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
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(proc.stdout)
    sys.stdout.flush()
    return proc.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--logn", type=int, default=16)
    ap.add_argument("--batch", type=int, default=4096, help="polynomials per GPU")
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
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args, sys.argv[1:]))

    # the host driver only supports dmabuf IPC: RCCL across processes needs this (already exported on the GPU boxes)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
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

    from ntt_aie_amd import _lib
    from ntt_aie_amd.dist import ShardedNTT

    logn, n, batch, p = args.logn, 1 << args.logn, args.batch, GOLDILOCKS
    eng = ShardedNTT(logn, p, g=7, word_bytes=8, device=local_rank)  # rank 0 makes the table, RCCL broadcast
    plan = eng.plan
    # rank r holds rows [r*batch, (r+1)*batch) of the job's [world*batch][N] input
    x = synth_batch(torch, batch, n, dev, first_row=rank * batch)
    y = torch.empty_like(x)
    stream = torch.cuda.current_stream()

    PREWARM = 8  # untimed, before the W warm-up steps: first touches of 4 GiB (TLB) and the clock ramp
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
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_ntt = batch * world * args.steps
    value = total_ntt / elapsed
    passes = plan.passes_for(batch)  # the decomposition the launcher picks for THIS batch (plan alternatives)
    if logn == 16 and batch == 4096:
        cfg_name = "BASELINE config 3's forward leg = the headline metric (N=2^16 Goldilocks, batch 4096 on one MI355X)" + (
            ", weak-scaled: 4096 per GPU" if world > 1 else "")
    elif logn == 16 and batch == 8192:
        cfg_name = ("BASELINE config 5 (N=2^16 Goldilocks, batch 65536 sharded across 8 MI355X = 8192 per GPU): %d GPU(s) x 8192 = %d "
                    "polynomials in this job" % (world, world * batch))
    else:
        cfg_name = "off-headline shape (N=2^%d, %d per GPU)" % (logn, batch)
    out = {
        "metric": "forward-NTT/s, N=2^%d 64-bit Goldilocks prime, batch=%d per GPU" % (logn, batch),
        "value": value, "unit": "NTT/s", "butterflies_per_s": value * (n // 2) * logn,
        # the reference's own operation count (profile/plot_efficiency.py:25,44: 5.5 * N * log2 N per transform)
        "ops_per_s_reference_convention": value * 5.5 * n * logn,
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_steps": PREWARM,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "%s; N=2^%d forward NTT, p=2^64-2^32+1, make_roots table g=7, batch=%d per GPU (%d in the job), "
                               "out-of-place, inputs resident in HBM, a[b][i] = splitmix64(0x9E3779B97F4A7C15 + b*N + i) mod p"
                               % (cfg_name, logn, batch, batch * world),
                   "baseline_config": (3 if (logn == 16 and batch == 4096) else 5 if (logn == 16 and batch == 8192) else None),
                   "batch_per_gpu": batch, "hbm_passes": len(passes),
                   "sharding": "contiguous batch rows per rank, no data-path collective",
                   "table_broadcast": (dist.get_backend() if use_dist else "none (single process)"),
                   "kernel_src_hash": _lib.kernel_source_hash()},
    }

    if rank == 0:
        src_hash = _lib.kernel_source_hash()
        # per-step distribution (median and min): one hipEvent pair per step on the launch stream
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(max(args.steps, 5))]
        for e0, e1 in evs:
            e0.record(stream)
            plan.forward(x, y, stream=stream)
            e1.record(stream)
        torch.cuda.synchronize()
        step_ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
        out["step_ms_median"] = step_ms[len(step_ms) // 2]
        out["step_ms_min"] = step_ms[0]
        # roofline: per-pass kernel durations from hipEvents on the launch stream
        reps = max(5, min(args.steps, 20))
        per_pass = np.zeros(len(passes))
        for _ in range(reps):
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
                    valu = valu_roofline(ent, passes, [float(v) for v in per_pass], batch, logn)
                else:
                    valu_src += "; not quoted: " + why
        out["roofline"] = {
            # The contract's figure: algorithmic HBM bytes / kernel time against the 8 TB/s spec peak (achieved, peak, unit, frac).
            # `bound` names the unit that actually binds this integer kernel: the vector ALU, whose own roofline
            # is the `valu` object; `frac_ceiling` is what `frac` could reach at most with this pass count.
            "bound": "valu",
            "bound_note": "achieved/peak/frac are the HBM roofline SURVEY 8(d) prescribes (algorithmic bytes over the spec peak); the "
                          "binding unit is the vector ALU under the 1400 W board power cap (roofline.valu: instructions per "
                          "butterfly against 1024 SIMDs x clock / 4 cycles; profiles/r02_power_probe.txt for the clock the cap allows) "
                          "-- each pass streams at ~0.87 of the device-copy rate, so HBM is the second constraint, not the first",
            "frac_ceiling": 1.0 / len(passes),
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "definition": "algorithmic bytes of one forward transform (2*N*8 B) x batch / summed duration of its "
                          "%d pass kernels (hipEvents on the launch stream); a %d-pass transform physically moves %dx its "
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
            out["cpu_baseline"] = cpu_baseline(logn, p, eng.table, threads=args.cpu_threads or None)
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
            torch.cuda.synchronize()
            inv_ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
            out["inverse"] = {"ms_per_step_median": inv_ms[len(inv_ms) // 2], "ms_per_step_min": inv_ms[0],
                              "NTT_per_s": batch / (inv_ms[len(inv_ms) // 2] * 1e-3),
                              "vs_forward_median": inv_ms[len(inv_ms) // 2] / out["step_ms_median"],
                              "round_trip_identical": bool(torch.equal(x2, x))}
            del x2
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
