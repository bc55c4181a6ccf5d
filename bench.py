#!/usr/bin/env python3
"""bench.py -- forward-NTT throughput of the MI355X engine on BASELINE.json's headline
configuration: N = 2^16, Goldilocks prime 2^64-2^32+1, batch 4096 per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one forward transform of the whole resident batch (input buffer -> output
buffer, both already in HBM).  Weak scaling: every rank owns `batch` polynomials and never
talks to the others in the timed region; the only collective is the one-off twiddle-table
broadcast from rank 0 (RCCL).  Rank 0 prints ONE JSON line.

Extra legs on rank 0 at N=1: per-pass kernel durations from hipEvents (roofline) and the CPU
baseline (the oracle restatement of src/test.cpp:34-60, timed on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GOLDILOCKS = 0xFFFFFFFF00000001
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s achievable


def synth_batch(torch, batch, n, device, seed):
    """Canonical residues covering (almost) the full range [0, p): hi word <= 2^32-2."""
    g = torch.Generator(device=device).manual_seed(seed)
    hi = torch.randint(0, 0xFFFFFFFF, (batch, n), dtype=torch.int64, device=device, generator=g)
    lo = torch.randint(0, 1 << 32, (batch, n), dtype=torch.int64, device=device, generator=g)
    return (hi << 32) | lo


def cpu_baseline(logn, p, table, cpu_seconds=20.0):
    """Oracle (port of the reference CPU verification path) on the host cores, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle_py as O

    n = 1 << logn
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, int(os.environ.get("NTT_BENCH_CPU_THREADS", "16")))  # the box's CPU share for one GPU
    rng = np.random.default_rng(1)
    probe = rng.integers(0, 2**63, size=(2, n), dtype=np.uint64)
    t0 = time.perf_counter()
    O.ntt(probe, table, p, nthreads=1)
    t1 = (time.perf_counter() - t0) / 2
    rate_1 = 1.0 / t1
    # bounded sample: about `cpu_seconds` of CPU work in total, spread over the host threads
    sample = int(max(cores * 2, min(16384, cpu_seconds / t1)))
    a = rng.integers(0, 2**63, size=(sample, n), dtype=np.uint64)
    t0 = time.perf_counter()
    O.ntt(a, table, p, nthreads=cores)
    tn = time.perf_counter() - t0
    rate_n = sample / tn
    best, used = (rate_n, cores) if rate_n >= rate_1 else (rate_1, 1)
    return {"value": best, "unit": "NTT/s", "cores": used, "kind": "port",
            "sample": "%d polynomials of N=2^%d on %d threads (%.2f s); 1-thread rate %.1f NTT/s"
                      % (sample, logn, cores, tn, rate_1),
            "value_1thread": rate_1, "butterflies_per_s": best * (n // 2) * logn}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--logn", type=int, default=16)
    ap.add_argument("--batch", type=int, default=4096, help="polynomials per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) even for one rank: exercises the twiddle broadcast path")
    args = ap.parse_args()

    # the host driver only supports dmabuf IPC: RCCL across processes needs this (already exported on the GPU boxes)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # rehearsal on a one-GPU box (tests only): NTT_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and
    # NTT_BENCH_BACKEND=gloo replaces RCCL, which refuses two ranks on one device
    if os.environ.get("NTT_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
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
    x = synth_batch(torch, batch, n, dev, seed=1234 + rank)
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
    out = {
        "metric": "forward-NTT/s, N=2^%d 64-bit Goldilocks prime, batch=%d per GPU" % (logn, batch),
        "value": value, "unit": "NTT/s", "butterflies_per_s": value * (n // 2) * logn,
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_steps": PREWARM,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "N=2^%d forward NTT, p=2^64-2^32+1, make_roots table g=7, batch=%d per GPU, "
                               "out-of-place, inputs resident in HBM" % (logn, batch),
                   "hbm_passes": plan.hbm_passes, "sharding": "batch rows, no data-path collective"},
    }

    if rank == 0:
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
        per_pass = np.zeros(plan.hbm_passes)
        for _ in range(reps):
            per_pass += np.array(plan.forward_profile(x, y, stream=stream))
        per_pass /= reps
        alg_bytes = 2.0 * n * 8 * batch  # 2*N*sizeof(word) per transform, read once + write once
        t_kernels = float(per_pass.sum()) * 1e-3
        achieved = alg_bytes / t_kernels / 1e9
        dom = int(per_pass.argmax())
        # HBM bytes per launch from the PMC counters (separate --pmc FETCH_SIZE / WRITE_SIZE runs of this
        # same command, FETCH_SIZE doubled per the gfx950 correction): profiles/r01_pmc_traffic.json
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        names = ["pass_%s_%d" % (kind, stages) for kind, _, stages in plan.passes]
        if os.path.exists(pmc) and logn == 16 and batch == 4096:
            k = json.load(open(pmc))["kernels"]
            if all(nm in k for nm in names):  # counters collected for this build's kernels
                traffic = sum(k[nm]["hbm_bytes_per_launch"] for nm in names)
                traffic_src = "profiles/r01_pmc_traffic.json"
        # the vector ALU's share of the kernel cycles from the SQ counters of the same command (profiles/r01_sq_counters.json):
        # for this integer workload the binding unit is the VALU, not HBM and not MFMA
        valu_busy = None
        sq = os.path.join(ROOT, "profiles", "r01_sq_counters.json")
        if os.path.exists(sq) and logn == 16 and batch == 4096:
            k = json.load(open(sq))["kernels"]
            if all(nm in k for nm in names):
                valu_busy = [k[nm]["valu_busy_frac_of_kernel"] for nm in names]
        out["roofline"] = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "definition": "algorithmic bytes of one forward transform (2*N*8 B) x batch / summed duration of its "
                          "%d pass kernels (hipEvents on the launch stream); traffic = PMC HBM bytes of the "
                          "same launches" % plan.hbm_passes,
            "algorithmic_bytes_per_transform": 2 * n * 8, "passes": plan.hbm_passes,
            "pass_stages": [stages for _, _, stages in plan.passes],
            "valu_busy_frac_per_pass": valu_busy, "valu_busy_source": "profiles/r01_sq_counters.json" if valu_busy else None,
            "pass_ms": [float(v) for v in per_pass], "dominant_pass": dom,
            # each pass kernel reads and writes every coefficient once: its own stream rate
            "pass_stream_GBs": [alg_bytes / (float(v) * 1e-3) / 1e9 for v in per_pass],
            "pass_stream_frac": [alg_bytes / (float(v) * 1e-3) / 1e9 / HBM_PEAK_GBS for v in per_pass],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(logn, p, eng.table)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
