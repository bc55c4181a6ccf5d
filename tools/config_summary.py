#!/usr/bin/env python3
"""Reduce the rocprofv3 counter CSVs of ONE BASELINE configuration (tools/config_profile.py <cfg> run under rocprofv3) to the
per-kernel and per-operation figures DESIGN.md section 4 quotes for the configurations other than the headline.

usage: config_summary.py pmc <cfg> <ops> <fetch.csv> <write.csv>   > profiles/rNN_<cfg>_pmc_traffic.json
       config_summary.py sq  <cfg> <ops> <sq.csv>                  > profiles/rNN_<cfg>_sq_counters.json
<ops> = operations the profiled program ran (its reps + 3 warm-up ones): every figure is per OPERATION (one forward
transform of the batch; one negacyclic product of the batch), because a product's kernels run a different number of times
per operation (the inverse column pass twice -- once per operand -- or once over both when the operands are contiguous).

Kernels are identified by tools/kernel_key.py (pass_kernel<PassCfg<...>, SC> and product_kernel<PassCfg<...>, ...>);
everything else the program launched (random fill, table generation) is ignored.  HBM bytes: FETCH_SIZE / WRITE_SIZE are KiB,
FETCH_SIZE doubled on gfx950 (tools/pmc_summary.py, tools/pmc_calib.hip), collected in separate passes as
MI355X_MICROARCH.md prescribes.  Stamped with the kernel-source hash."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from configs import CONFIGS, algorithmic_bytes, butterflies
from hw import SIMDS, valu_frac_of_peak
from kernel_key import parse_kernel


def transforms_per_op(cfg: dict, pk: dict) -> int:
    """How many LOG_M-stage networks per polynomial of the batch this kernel runs in ONE operation."""
    if pk["kind"] == "product":
        return 3                      # inverse of a, inverse of b, forward of the product
    if cfg["op"] == "polymul":
        return 2 if pk["inv"] else 1  # both operands go through the inverse column passes, the product through the forward ones
    return 1


def kernel_butterflies_per_op(cfg: dict, pk: dict) -> float:
    return transforms_per_op(cfg, pk) * cfg["batch"] * (1 << (cfg["logn"] - 1)) * pk["log_m"]


def rows_by_kernel(path, min_grid_frac=0.25):
    """{key: {"pk": parsed, counter: [values], "_dur_ns": [...]}} for the launches of the timed batch (a kernel's launches with a
    grid below a quarter of its largest are parity-sized warm-ups of some other shape: dropped)."""
    raw = collections.defaultdict(list)
    for r in csv.DictReader(open(path, newline="")):
        pk = parse_kernel(r["Kernel_Name"])
        if pk is not None:
            raw[pk["key"]].append((pk, r))
    out = {}
    for key, rows in raw.items():
        gmax = max(int(r["Grid_Size"]) for _, r in rows)
        d = {"pk": rows[0][0], "_dur_ns": [], "_dispatches": set()}
        for _, r in rows:
            if int(r["Grid_Size"]) < min_grid_frac * gmax:
                continue
            d.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            disp = r.get("Dispatch_Id")
            if disp not in d["_dispatches"]:
                d["_dispatches"].add(disp)
                if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                    d["_dur_ns"].append(float(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        out[key] = d
    return out


def head(cfg_key, ops, src_hash):
    c = CONFIGS[cfg_key]
    return {"src_hash": src_hash, "config": cfg_key, "name": c["name"], "operations_profiled": ops,
            "algorithmic_bytes_per_op": algorithmic_bytes(c), "butterflies_per_op": butterflies(c)}


def summarize_pmc(cfg_key, ops, fetch_csv, write_csv, src_hash=None):
    c = CONFIGS[cfg_key]
    fetch, write = rows_by_kernel(fetch_csv), rows_by_kernel(write_csv)
    out = head(cfg_key, ops, src_hash)
    out["note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of tools/config_profile.py; KiB; FETCH_SIZE "
                   "doubled (gfx950), WRITE_SIZE exact; per kernel: mean bytes per launch and launches per operation; per_op sums "
                   "launches x bytes over the kernels of one operation")
    out["kernels"] = {}
    tot = 0.0
    for key, f in fetch.items():
        if key not in write or "FETCH_SIZE" not in f or "WRITE_SIZE" not in write[key]:
            continue
        fv, wv = f["FETCH_SIZE"], write[key]["WRITE_SIZE"]
        fb, wb = 2.0 * 1024.0 * sum(fv) / len(fv), 1024.0 * sum(wv) / len(wv)
        per_op = len(fv) / float(ops)
        out["kernels"][key] = {"short": f["pk"]["short"], "kind": f["pk"]["kind"], "launches": len(fv), "launches_per_op": per_op,
                               "fetch_bytes_corrected": fb, "write_bytes": wb, "hbm_bytes_per_launch": fb + wb,
                               "hbm_bytes_per_op": (fb + wb) * per_op}
        tot += (fb + wb) * per_op
    out["per_op"] = {"hbm_bytes": tot, "ratio_to_algorithmic": tot / algorithmic_bytes(c)}
    return out


def summarize_sq(cfg_key, ops, sq_csv, src_hash=None):
    c = CONFIGS[cfg_key]
    sq = rows_by_kernel(sq_csv)
    out = head(cfg_key, ops, src_hash)
    out["note"] = ("rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU "
                   "SQ_INSTS_SALU GRBM_GUI_ACTIVE of tools/config_profile.py; per kernel, means per launch; valu_instr_per_butterfly = "
                   "SQ_INSTS_VALU per operation / (butterflies this kernel runs per operation / 64): the product kernel runs three "
                   "LOG_M-stage networks per polynomial and its two word-by-word products count as overhead on those; SQ_* cycle "
                   "counters are quad-cycles summed over waves; kernel_cycles = GRBM_GUI_ACTIVE / 8 XCDs; valu_frac_of_peak_all_vop3 prices every VALU "
                   "instruction at the 4 cycles the VOP3-class forms take (measured: tools/hw.py; plain moves / adds cost 2: an upper estimate, "
                   "bench.config_roofline refines it with the statement's mix)")
    out["kernels"] = {}
    tot_valu = tot_cyc = tot_bf = 0.0
    for key, d in sq.items():
        pk = d["pk"]
        if "SQ_INSTS_VALU" not in d:
            continue
        n = len(d["SQ_INSTS_VALU"])
        m = {k: sum(v) / len(v) for k, v in d.items() if not k.startswith("_") and k != "pk"}
        per_op = n / float(ops)
        bf_op = kernel_butterflies_per_op(c, pk)
        e = {"short": pk["short"], "kind": pk["kind"], "direction": "inv" if pk["inv"] else "fwd", "log_m": pk["log_m"],
             "launches": n, "launches_per_op": per_op, "butterflies_per_op": bf_op}
        e.update(m)
        e["valu_instr_per_butterfly"] = m["SQ_INSTS_VALU"] * per_op / (bf_op / 64.0)
        if "GRBM_GUI_ACTIVE" in m:
            cyc = m["GRBM_GUI_ACTIVE"] / 8
            e["kernel_cycles"] = cyc
            if d["_dur_ns"]:
                dur = sum(d["_dur_ns"]) / len(d["_dur_ns"])
                e["duration_us"] = dur / 1e3
                e["held_clock_GHz"] = cyc / dur
            if "SQ_WAVE_CYCLES" in m:
                e["mean_waves_per_simd"] = m["SQ_WAVE_CYCLES"] * 4 / cyc / SIMDS
                if "SQ_WAIT_INST_ANY" in m:
                    e["wave_issue_stall_frac"] = m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]
                if "SQ_WAIT_ANY" in m:
                    e["wave_parked_frac"] = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]
            e["valu_frac_of_peak_all_vop3"] = valu_frac_of_peak(m["SQ_INSTS_VALU"], cyc)
            tot_cyc += cyc * per_op
        tot_valu += m["SQ_INSTS_VALU"] * per_op
        tot_bf += bf_op
        out["kernels"][key] = e
    out["per_op"] = {"valu_instr": tot_valu, "butterflies_in_profiled_kernels": tot_bf,
                     "valu_instr_per_butterfly": tot_valu / (tot_bf / 64.0) if tot_bf else None,
                     "kernel_cycles": tot_cyc,
                     "valu_frac_of_peak_all_vop3": valu_frac_of_peak(tot_valu, tot_cyc) if tot_cyc else None}
    return out


def main():
    from ntt_aie_amd._lib import kernel_source_hash

    mode, cfg_key, ops = sys.argv[1], sys.argv[2], int(sys.argv[3])
    if mode == "pmc":
        out = summarize_pmc(cfg_key, ops, sys.argv[4], sys.argv[5], src_hash=kernel_source_hash())
    elif mode == "sq":
        out = summarize_sq(cfg_key, ops, sys.argv[4], src_hash=kernel_source_hash())
    else:
        raise SystemExit(__doc__)
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
