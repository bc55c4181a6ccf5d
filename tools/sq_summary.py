#!/usr/bin/env python3
"""Reduce a rocprofv3 --pmc SQ_* GRBM_GUI_ACTIVE counter_collection.csv of the bench command to per-kernel figures:
VALU instructions per butterfly, mean waves per SIMD, VALU instruction count x 4 cycles over the kernel cycles, stall split.
usage: sq_summary.py counter_collection.csv > profiles/rNN_sq_counters.json"""
import collections, csv, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ntt_aie_amd._lib import kernel_source_hash  # the kernels these counters belong to (bench.py checks it)

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "pass_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 1000000:
        cfg = r["Kernel_Name"].split("PassCfg<")[1].split(">")[0].split(",")
        acc["pass_%s_%s" % ("contig" if cfg[3].strip() == "true" else "col", cfg[1].strip())][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"src_hash": kernel_source_hash(), "note": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU "
               "SQ_INSTS_VALU GRBM_GUI_ACTIVE -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline (one pass, 8 SQ slots); SQ_* cycle "
               "counters are quad-cycles summed over waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs; means over the batch-4096 launches "
               "(N = 2^16: 4096 * 32768 * 8 butterflies per launch, 1024 SIMDs); valu_instr_x4cyc_over_kernel_cycles = SQ_INSTS_VALU x an ASSUMED 4 cycles per wave-instruction / (1024 SIMDs x kernel cycles): an instruction-count estimate, not a busy-cycle measurement (SQ_ACTIVE_INST_VALU equals SQ_INSTS_VALU on this counter set)", "kernels": {}}
for k, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    m.update(kernel_cycles=cyc, mean_waves_per_simd=m["SQ_WAVE_CYCLES"] * 4 / cyc / 1024,
             valu_instr_x4cyc_over_kernel_cycles=m["SQ_ACTIVE_INST_VALU"] * 4 / cyc / 1024,
             wave_parked_frac=m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], wave_issue_stall_frac=m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"],
             valu_instr_per_butterfly=m["SQ_INSTS_VALU"] / (4096 * 32768 * 8 / 64))
    out["kernels"][k] = m
json.dump(out, sys.stdout, indent=1)
print()
