#!/usr/bin/env python3
"""Reduce a rocprofv3 --pmc SQ_* GRBM_GUI_ACTIVE counter_collection.csv of the bench command to per-kernel figures:
VALU instructions per butterfly, mean waves per SIMD, VALU instruction count x 4 cycles (the measured throughput of the VOP3-class forms, tools/hw.py) over the kernel cycles, stall split,
launch duration and the shader clock the kernel held (GRBM_GUI_ACTIVE / 8 XCDs / duration).
usage: sq_summary.py counter_collection.csv [--batch 4096] [--logn 16] > profiles/rNN_sq_counters.json

Kernels are keyed by their FULL PassCfg<...> argument list (tools/kernel_key.py): the forward and the inverse kernel of one
pass shape are two entries, never averaged together."""
import collections, csv, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hw import SIMDS, VALU_PEAK_CYCLES_VOP3, valu_frac_of_peak
from kernel_key import parse_pass_kernel


def summarize(path, batch=4096, logn=16, min_grid=1000000, src_hash=None):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for r in csv.DictReader(open(path, newline="")):
        pk = parse_pass_kernel(r["Kernel_Name"])
        if pk is None or int(r["Grid_Size"]) < min_grid:
            continue
        acc[pk["key"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r.get("Start_Timestamp") and r.get("End_Timestamp"):
            acc[pk["key"]]["_dur_ns"].append(float(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        meta[pk["key"]] = pk
    out = {"src_hash": src_hash,
           "note": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU "
                   "SQ_INSTS_VALU GRBM_GUI_ACTIVE -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-valu-floor (one pass, 8 SQ "
                   "slots); SQ_* cycle counters are quad-cycles summed over waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs; means over the "
                   "batch-%d launches (N = 2^%d: batch * N/2 * LOG_M butterflies per launch, %d SIMDs); one entry per kernel instantiation, "
                   "keyed by the full PassCfg<...> argument list (INV = 5th argument); valu_frac_of_peak_all_vop3 = SQ_INSTS_VALU x %g cycles per "
                   "wave-instruction (the MEASURED throughput of the VOP3-class forms, tools/hw.py; plain moves / adds cost 2, so this is an "
                   "UPPER estimate: bench.py refines it with the statement's mix) / (%d SIMDs x kernel cycles), clock-free (SQ_ACTIVE_INST_VALU "
                   "equals SQ_INSTS_VALU on this counter set); "
                   "held_clock_GHz = GRBM_GUI_ACTIVE / 8 / launch duration under the profiler"
                   % (batch, logn, SIMDS, VALU_PEAK_CYCLES_VOP3, SIMDS),
           "batch": batch, "logn": logn, "kernels": {}}
    for k, v in acc.items():
        m = {c: sum(x) / len(x) for c, x in v.items() if not c.startswith("_")}
        pk = meta[k]
        wave_butterflies = batch * (1 << (logn - 1)) * pk["log_m"] / 64.0
        m.update(short=pk["short"], direction="inv" if pk["inv"] else "fwd", launches=len(v["SQ_INSTS_VALU"]) if "SQ_INSTS_VALU" in v else 0)
        if "GRBM_GUI_ACTIVE" in m:
            cyc = m["GRBM_GUI_ACTIVE"] / 8
            m["kernel_cycles"] = cyc
            if v.get("_dur_ns"):
                dur = sum(v["_dur_ns"]) / len(v["_dur_ns"])
                m["duration_us"] = dur / 1e3
                m["held_clock_GHz"] = cyc / dur
            if "SQ_WAVE_CYCLES" in m:
                m["mean_waves_per_simd"] = m["SQ_WAVE_CYCLES"] * 4 / cyc / SIMDS
            if "SQ_INSTS_VALU" in m:
                m["valu_frac_of_peak_all_vop3"] = valu_frac_of_peak(m["SQ_INSTS_VALU"], cyc)
        if "SQ_WAVE_CYCLES" in m:
            if "SQ_WAIT_ANY" in m:
                m["wave_parked_frac"] = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]
            if "SQ_WAIT_INST_ANY" in m:
                m["wave_issue_stall_frac"] = m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]
        if "SQ_INSTS_VALU" in m:
            m["valu_instr_per_butterfly"] = m["SQ_INSTS_VALU"] / wave_butterflies
        out["kernels"][k] = m
    return out


def main():
    from ntt_aie_amd._lib import kernel_source_hash  # the kernels these counters belong to (bench.py checks it)

    a = sys.argv[1:]
    batch = int(a[a.index("--batch") + 1]) if "--batch" in a else 4096
    logn = int(a[a.index("--logn") + 1]) if "--logn" in a else 16
    json.dump(summarize(a[0], batch, logn, src_hash=kernel_source_hash()), sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
