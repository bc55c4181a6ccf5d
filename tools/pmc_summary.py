#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection.csv files (separate passes, as
MI355X_MICROARCH.md's HBM section prescribes) to HBM bytes per launch of each pass kernel.

usage: pmc_summary.py <bench_fetch.csv> <bench_write.csv> [<calib_fetch.csv> <calib_write.csv>] > profiles/rNN_pmc_traffic.json

Units: both counters are KiB.  gfx950 correction: FETCH_SIZE counts half of the bytes read (the calibration
copies of tools/pmc_calib.hip show it: 1 GiB copied reads 0.5 GiB by the counter), so it is doubled;
WRITE_SIZE is exact.

Kernels are keyed by their FULL PassCfg<...> argument list (tools/kernel_key.py): the forward and the inverse kernel of one
pass shape are two entries."""
import csv, json, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_key import parse_pass_kernel


def per_kernel(path, counter):
    acc = defaultdict(list)
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def short(name):
    return name.split("(")[0].split("::")[-1].split("<")[0]


def summarize(fetch_csv, write_csv, calib_fetch=None, calib_write=None, src_hash=None):
    fetch, write = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    out = {"src_hash": src_hash,
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; unit KiB; FETCH_SIZE doubled "
                   "(gfx950 correction, confirmed by tools/pmc_calib.hip), WRITE_SIZE exact; launches with the "
                   "largest grid of each kernel only (the timed batch, not the parity smoke); one entry per kernel "
                   "instantiation, keyed by the full PassCfg<...> argument list (INV = 5th argument)",
           "kernels": {}}
    if calib_fetch and calib_write:
        cf, cw = per_kernel(calib_fetch, "FETCH_SIZE"), per_kernel(calib_write, "WRITE_SIZE")
        out["calibration"] = {**{f"{short(k)} FETCH_SIZE_KiB_for_1GiB": sum(v) / len(v) for k, v in cf.items()},
                              **{f"{short(k)} WRITE_SIZE_KiB_for_1GiB": sum(v) / len(v) for k, v in cw.items()}}
    for k in fetch:
        pk = parse_pass_kernel(k)
        if pk is None or k not in write:
            continue
        f_big = [v for v in fetch[k] if v > 0.5 * max(fetch[k])]
        w_big = [v for v in write[k] if v > 0.5 * max(write[k])]
        fb, wb = 2.0 * 1024.0 * sum(f_big) / len(f_big), 1024.0 * sum(w_big) / len(w_big)
        out["kernels"][pk["key"]] = {"short": pk["short"], "direction": "inv" if pk["inv"] else "fwd", "launches": len(f_big),
                                     "FETCH_SIZE_KiB_raw": sum(f_big) / len(f_big), "fetch_bytes_corrected": fb, "write_bytes": wb,
                                     "hbm_bytes_per_launch": fb + wb}
    return out


def main():
    from ntt_aie_amd._lib import kernel_source_hash  # the kernels these counters belong to (bench.py checks it)

    out = summarize(sys.argv[1], sys.argv[2], *(sys.argv[3:5] if len(sys.argv) >= 5 else ()), src_hash=kernel_source_hash())
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
