#!/usr/bin/env python3
"""Comparison plots in the reference's conventions (profile/plot_efficiency.py:25-27,44-46; plot_exectime.py:27-29):
efficiency = 5.5*N*log2(N) operations / kernel time / peak, for the MI355X series of profiles/kerneltime/ beside the
reference's recorded AIE and A100 series (their published numbers, copied below as data), and launch-to-completion time
of the reference's 10-launch procedure beside the AIE's 16-tile series.

usage: plot_profiles.py [profiles_dir]   -> profiles/efficiency_mi355x.{png,csv}, profiles/kerneltime_mi355x.png, profiles/exectime_mi355x.png"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from hw import PEAK_GOPS_INT32  # noqa: E402

# the reference's recorded series (profile/kerneltime/aie.csv, gpu.csv): N, kernel microseconds
AIE_KERNEL_US = {512: 8.86256, 1024: 10.67568, 2048: 14.3748, 4096: 22.06464}
A100_KERNEL_US = {256: 12.004, 512: 13.497, 1024: 16.365, 2048: 21.510, 4096: 19.276, 8192: 21.179, 16384: 24.203,
                  32768: 31.337, 65536: 45.942, 131072: 81.350}
# peaks the reference divides by (profile/plot_efficiency.py:27, 46): A100 4280 GOPS, AIE 88 GOPS
PEAK_GOPS = {"aie": 88.0, "a100": 4280.0,
             # MI355X: 32-bit integer vector rate, 256 CUs x 4 SIMD-32 x 32 lanes per clock x 2.4 GHz = 78.6 TOPS (tools/hw.py;
             # MI355X_MICROARCH.md: a wave64 instruction takes 2 cycles of a SIMD's throughput).  Rounds 1-4 divided by a 16-lane
             # figure -- what ONE wave alone sustains -- and published efficiencies above 1; a stated peak must not be exceeded.
             "mi355x": PEAK_GOPS_INT32}
# the reference's 16-tile launch-to-completion series, trimmed means of profile/exectime/ntt_16core_logn*.csv (plot_exectime.py rule)
AIE16_EXEC_US = {256: 288.4, 512: 261.8, 1024: 274.8, 2048: 279.4, 4096: 288.0, 8192: 319.4}


def read_rows(path):
    out = {}
    for line in open(path):
        if line.strip():
            n, us = line.split(",")[:2]
            out[int(n)] = float(us)
    return out


def main():
    from ntt_aie_amd.host import efficiency, trimmed_mean  # host-side formatters only (no GPU needed)

    prof = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles")
    mi1 = read_rows(os.path.join(prof, "kerneltime", "mi355x.csv"))
    mib = read_rows(os.path.join(prof, "kerneltime", "mi355x_batch.csv"))
    rp = os.path.join(prof, "kerneltime", "mi355x_rocprof.csv")  # true kernel durations (rocprofv3), when collected
    mir = read_rows(rp) if os.path.exists(rp) else {}
    series = {"Ryzen AI Engine (reference)": (AIE_KERNEL_US, PEAK_GOPS["aie"]),
              "NVIDIA A100 (reference)": (A100_KERNEL_US, PEAK_GOPS["a100"]),
              "MI355X, batch 1 (hipEvents around the launch)": (mi1, PEAK_GOPS["mi355x"]),
              "MI355X, saturating batch (per transform)": (mib, PEAK_GOPS["mi355x"])}
    if mir:
        series["MI355X, batch 1 (rocprofv3 kernel time)"] = (mir, PEAK_GOPS["mi355x"])
    with open(os.path.join(prof, "efficiency_mi355x.csv"), "w") as f:
        f.write("series,N,kernel_us,gops,efficiency\n")
        for name, (rows, peak) in series.items():
            for n, us in sorted(rows.items()):
                eff = efficiency(n, us, peak)
                f.write("%s,%d,%.5f,%.2f,%.5f\n" % (name, n, us, eff * peak, eff))
    exe = {}
    for logn in range(8, 14):
        p = os.path.join(prof, "exectime", "ntt_mi355x_logn%d.csv" % logn)
        if os.path.exists(p):
            exe[1 << logn] = trimmed_mean([float(v) for v in open(p).read().split()])
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except ImportError:
        print("matplotlib missing: wrote the CSV only")
        return
    fig, ax = plt.subplots(figsize=(10, 6))
    for name, (rows, peak) in series.items():
        ns = sorted(rows)
        ax.plot(ns, [efficiency(n, rows[n], peak) for n in ns], marker="o", label=name)
    ax.set_xscale("log", base=2)
    ax.set_yscale("log")
    ax.set_xlabel("Data size")
    ax.set_ylabel("Efficiency (5.5 N log2 N ops / time / peak)")
    ax.grid(True)
    ax.legend()
    fig.savefig(os.path.join(prof, "efficiency_mi355x.png"), dpi=120, bbox_inches="tight")
    # profile/plot_kerneltime.py: kernel microseconds against the data size
    fig, ax = plt.subplots(figsize=(10, 6))
    for name, (rows, _) in series.items():
        if "saturating" in name:
            continue
        ns = sorted(rows)
        ax.plot(ns, [rows[n] for n in ns], marker="o", label=name)
    ax.set_xscale("log", base=2)
    ax.set_xlabel("Data size")
    ax.set_ylabel("Kernel Time (us)")
    ax.grid(True)
    ax.legend()
    fig.savefig(os.path.join(prof, "kerneltime_mi355x.png"), dpi=120, bbox_inches="tight")
    fig, ax = plt.subplots(figsize=(10, 6))
    ax.plot(sorted(AIE16_EXEC_US), [AIE16_EXEC_US[n] for n in sorted(AIE16_EXEC_US)], marker="o", label="AIE, 16 tiles (reference)")
    if exe:
        ax.plot(sorted(exe), [exe[n] for n in sorted(exe)], marker="o", label="MI355X (launch + wait)")
    ax.set_xscale("log", base=2)
    ax.set_xlabel("Data size")
    ax.set_ylabel("Execution Time (us)")
    ax.grid(True)
    ax.legend()
    fig.savefig(os.path.join(prof, "exectime_mi355x.png"), dpi=120, bbox_inches="tight")
    print("wrote efficiency_mi355x.{csv,png}, kerneltime_mi355x.png, exectime_mi355x.png under", prof)


if __name__ == "__main__":
    main()
