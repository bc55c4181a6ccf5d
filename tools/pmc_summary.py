#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection.csv files (separate passes, as
MI355X_MICROARCH.md's HBM section prescribes) to HBM bytes per launch of each pass kernel.

usage: pmc_summary.py <bench_fetch.csv> <bench_write.csv> [<calib_fetch.csv> <calib_write.csv>] > profiles/rNN_pmc_traffic.json

Units: both counters are KiB.  gfx950 correction: FETCH_SIZE counts half of the bytes read (the calibration
copies of tools/pmc_calib.hip show it: 1 GiB copied reads 0.5 GiB by the counter), so it is doubled;
WRITE_SIZE is exact."""
import csv, json, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ntt_aie_amd._lib import kernel_source_hash  # the kernels these counters belong to (bench.py checks it)


def per_kernel(path, counter):
    acc = defaultdict(list)
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def short(name):
    if "pass_kernel" in name:
        cfg = name.split("PassCfg<")[1].split(">")[0].split(",")
        return f"pass_{'contig' if cfg[3].strip() == 'true' else 'col'}_{cfg[1].strip()}"
    return name.split("(")[0].split("::")[-1].split("<")[0]


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"src_hash": kernel_source_hash(),
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; unit KiB; FETCH_SIZE doubled "
                   "(gfx950 correction, confirmed by tools/pmc_calib.hip), WRITE_SIZE exact; launches with the "
                   "largest grid of each kernel only (the timed batch, not the parity smoke)",
           "kernels": {}}
    if len(sys.argv) >= 5:
        cf, cw = per_kernel(sys.argv[3], "FETCH_SIZE"), per_kernel(sys.argv[4], "WRITE_SIZE")
        out["calibration"] = {**{f"{short(k)} FETCH_SIZE_KiB_for_1GiB": sum(v) / len(v) for k, v in cf.items()},
                              **{f"{short(k)} WRITE_SIZE_KiB_for_1GiB": sum(v) / len(v) for k, v in cw.items()}}
    for k in fetch:
        if "pass_kernel" not in k or k not in write:
            continue
        f_big = [v for v in fetch[k] if v > 0.5 * max(fetch[k])]
        w_big = [v for v in write[k] if v > 0.5 * max(write[k])]
        fb, wb = 2.0 * 1024.0 * sum(f_big) / len(f_big), 1024.0 * sum(w_big) / len(w_big)
        out["kernels"][short(k)] = {"kernel": k[:160], "launches": len(f_big), "FETCH_SIZE_KiB_raw": sum(f_big) / len(f_big),
                                    "fetch_bytes_corrected": fb, "write_bytes": wb, "hbm_bytes_per_launch": fb + wb}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
