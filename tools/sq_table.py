#!/usr/bin/env python3
"""Per-kernel means of every counter in a rocprofv3 --pmc counter_collection.csv, with the dispatch duration and the
effective clock (GRBM_GUI_ACTIVE / 8 XCDs / duration) when that counter was collected.  Only launches whose grid is at
least --min-grid work-items (the timed batch, not the warm-ups of tiny sizes).
usage: sq_table.py counter_collection.csv [--min-grid 1000000]"""
import collections
import csv
import sys

path = sys.argv[1]
min_grid = int(sys.argv[sys.argv.index("--min-grid") + 1]) if "--min-grid" in sys.argv else 1000000
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
    if "pass_kernel" not in r["Kernel_Name"] or int(r["Grid_Size"]) < min_grid:
        continue
    cfg = r["Kernel_Name"].split("PassCfg<")[1].split(">")[0].replace(" ", "")
    key = "%s vgpr=%s lds=%s" % (cfg, r["VGPR_Count"], r["LDS_Block_Size"])
    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    acc[key]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for key, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    print(key, "launches", len(v["_dur_us"]) // max(1, len(v) - 1))
    line = "   dur %.1f us" % m["_dur_us"]
    if "GRBM_GUI_ACTIVE" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8
        line += "  kernel_cycles %.0f  clock %.3f GHz" % (cyc, cyc / m["_dur_us"] / 1e3)
        if "SQ_WAVE_CYCLES" in m:
            line += "  waves/SIMD %.2f" % (m["SQ_WAVE_CYCLES"] * 4 / cyc / 1024)
        if "SQ_INSTS_VALU" in m:
            line += "  VALU x 4 cycles / kernel cycles (measured VOP3-class throughput) %.3f" % (m["SQ_INSTS_VALU"] * 4 / cyc / 1024)
    print(line)
    if "SQ_WAVE_CYCLES" in m:
        wc = m["SQ_WAVE_CYCLES"]
        print("   of wave-cycles: " + "  ".join("%s %.3f" % (c, m[c] / wc) for c in sorted(m) if c.startswith("SQ_") and c != "SQ_WAVE_CYCLES" and ("WAIT" in c or "ACTIVE" in c or "CYCLES" in c)))
    print("   raw: " + "  ".join("%s %.4g" % (c, m[c]) for c in sorted(m) if not c.startswith("_")))
