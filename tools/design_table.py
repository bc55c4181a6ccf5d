#!/usr/bin/env python3
"""Rewrite the generated tables of DESIGN.md section 4 from the committed profile set profiles/<round>_*: the headline table
(bench.py's line, rocprofv3 kernel stats, counters, power probe) and the three-row table that puts BASELINE configs 2 / 3 / 4
under one evidence standard (tools/bench_configs.py's roofline objects + tools/config_summary.py's counter summaries), plus the
FieldM64 ratio of section 3.3.  The tables live between <!-- generated:NAME --> ... <!-- /generated:NAME --> markers, so the
text can never drift from the files it cites.  usage: python tools/design_table.py [spread text]"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (PROFILE_ROUND: the one place the round tag lives)

R = bench.PROFILE_ROUND


def P(f):
    return os.path.join(ROOT, "profiles", f)


def jl(path):
    return [json.loads(l) for l in open(path) if l.startswith("{")]


def power(tag, text):
    m = re.search(re.escape(tag) + r"\s+([\d.]+) ms/iter  power W: max [\d.]+ mean ([\d.]+).*?sclk MHz: \[([\d, ]+)\]", text)
    cl = sorted(int(x) for x in m.group(3).split(","))
    return float(m.group(2)), cl[len(cl) // 2] / 1000


def kernel_avg_ms(stats_csv, needle):
    for r in csv.DictReader(open(stats_csv)):
        if needle(r["Name"]):
            return float(r["AverageNs"]) / 1e6
    return float("nan")


def headline(spread):
    d = jl(P("%s_bench.json" % R))[-1]
    r, v = d["roofline"], d["roofline"]["valu"]
    st = P("%s_kernel_stats.csv" % R)
    k0 = kernel_avg_ms(st, lambda n: "pass_kernel" in n and "FieldGL, 8, 0, true, false" in n)
    k1 = kernel_avg_ms(st, lambda n: "pass_kernel" in n and "FieldGL, 8, 4, false, false" in n)
    pp = open(P("%s_power_probe.txt" % R)).read()
    pt, ct = power("forward (real)", pp)
    pa, ca = power("forward (L2 loads, no stores)", pp)
    pc, _ = power("copy (xor kernel)", pp)
    cb = d["cpu_baseline"]
    w = v.get("frac_at_held_clock_weighted")
    im = v.get("issue_model") or {}
    cc = im.get("class_cycles", {})
    rk = d["ranks"][0]
    return """| quantity (`profiles/{R}_bench.json`, `{R}_kernel_stats.csv`, `{R}_pmc_traffic.json`, `{R}_sq_counters.json`) | value |
|---|---|
| throughput | **{val:.2f} M NTT/s = {bf:.2f}e12 butterflies/s, {ms:.3f} ms per step** ({spread}) |
| verification in the same run | rank 0 on `{bus}`: inverse(forward(x)) == x over the whole shard, out[b][0] == Σ a[b][:] mod p on {rows} rows: `all_ranks_verified` {ver} |
| pass kernels, hipEvents / rocprofv3 | CONTIG {p0:.3f} / {k0:.3f} ms, column {p1:.3f} / {k1:.3f} ms |
| `roofline.achieved / peak / frac` (contract: algorithmic bytes over 8 TB/s) | {ach:.2f} TB/s / 8 TB/s = **{frac:.3f}**; `frac_step` (over the line's own `ms_per_step`) {fs:.3f}; `frac_ceiling` 0.5 (two passes) |
| `roofline.traffic` (PMC, forward kernels) | {tr:.3f} GB = {trr:.2f} × algorithmic: two trips, no over-fetch within a pass |
| each pass's stream rate | {s0:.2f} / {s1:.2f} TB/s = {f0:.2f}–{f1:.2f} of peak = {c0:.2f}–{c1:.2f} of the same-process device copy ({cp:.2f} TB/s) |
| **`roofline.bound` = `valu`**, flat price (every VALU form 4 cycles) | {ipb:.2f} VALU per butterfly ({i0:.2f} / {i1:.2f}); peak = 1024 SIMDs·f/(4·instr)·64 = {pk:.2f}e12 bf/s at 2.4 GHz; frac {f24:.2f} at 2.4 GHz, {fh:.2f} at the held clock ({h0:.2f}–{h1:.2f} GHz under the counter run) |
| … priced with **measured issue costs** (`{R}_valu_issue_cost.json` at 4 waves per SIMD, `{R}_valu_mix.json`) | carry / compare / `v_mad_u64_u32` forms {cy:.2f} cycles per wave-instruction, plain moves {pl:.2f}; a butterfly {w0:.1f} / {w1:.1f} cycles ⇒ **{w:.2f} of the issue capacity at the held clock** ({wv}) |
| what holds the clock | the 1400 W board cap: transform {pt:.0f} W at {ct:.2f} GHz; arithmetic alone {pa:.0f} W at {ca:.2f} GHz; copy alone {pc:.0f} W (`profiles/{R}_power_probe.txt`) |
| VALU floor (same kernels, loads from L2, no stores; measured in the bench run) | {fl0:.3f} + {fl1:.3f} ms (cycle view: `{R}_sq_real_vs_floor.txt`) |
| CPU baseline (oracle port, same run, **the GPU's own input rows**) | {c1t:.0f} NTT/s on 1 thread; {call:.2f} k on the {cores} cores the box's cgroup quota allows (affinity mask {aff}); {crows} rows |
| inverse (config 3's second leg) | {inv:.2f} ms, {invr:.3f} × forward, round trip identical |
""".format(R=R, val=d["value"] / 1e6, bf=d["butterflies_per_s"] / 1e12, ms=d["ms_per_step"], spread=spread,
           bus=rk.get("pci_bus_id") or rk.get("uuid"), rows=rk["rows_sampled"], ver=str(d["all_ranks_verified"]).lower(),
           p0=r["pass_ms"][0], k0=k0, p1=r["pass_ms"][1], k1=k1, ach=r["achieved"] / 1e3, frac=r["frac"], fs=r["frac_step"],
           tr=r["traffic"] / 1e9, trr=r["traffic"] / r["algorithmic_bytes_per_launch"],
           s0=r["pass_stream_GBs"][0] / 1e3, s1=r["pass_stream_GBs"][1] / 1e3, f0=min(r["pass_stream_frac"]), f1=max(r["pass_stream_frac"]),
           c0=min(r["pass_stream_frac_of_device_copy"]), c1=max(r["pass_stream_frac_of_device_copy"]), cp=r["device_copy"]["GBs"] / 1e3,
           ipb=v["instr_per_butterfly_mean"], i0=v["instr_per_butterfly"][0], i1=v["instr_per_butterfly"][1],
           pk=v["peak_butterflies_per_s"] / 1e12, f24=v["frac_at_2.4GHz"], fh=v["frac_at_held_clock"],
           h0=min(v["held_clock_GHz"]), h1=max(v["held_clock_GHz"]),
           cy=cc.get("carry", float("nan")), pl=cc.get("plain", float("nan")),
           w0=(v.get("issue_cycles_per_butterfly_weighted") or [float("nan")] * 2)[0],
           w1=(v.get("issue_cycles_per_butterfly_weighted") or [float("nan")] * 2)[1],
           w=w if w is not None else float("nan"), wv="saturated" if v.get("saturated") else "not saturated: see the reading below",
           pt=pt, ct=ct, pa=pa, ca=ca, pc=pc, fl0=r["valu_floor_pass_ms"][0], fl1=r["valu_floor_pass_ms"][1],
           c1t=cb["value_1thread"], call=cb["value"] / 1e3, cores=cb["cores"], aff=cb["host_affinity_cores"], crows=cb.get("sample_rows", 0),
           inv=d["inverse"]["ms_per_step_median"], invr=d["inverse"]["vs_forward_median"])


def configs():
    """configs 2 / 3 / 4: one row each, the same columns."""
    rows = {c["config"].split(":")[0]: c for c in jl(P("%s_bench_all_configs.jsonl" % R))}
    head = jl(P("%s_bench.json" % R))[-1]
    out = ["| config | time per operation | `frac` of 8 TB/s (ceiling) | HBM bytes, PMC (÷ algorithmic) | VALU per butterfly; ×4 cycles ÷ kernel cycles | "
           "waves / SIMD; held clock | binds |", "|---|---|---|---|---|---|---|"]

    def row(name, c, extra=""):
        r = c["roofline"]
        v = r.get("valu") or {}
        ks = list((v.get("kernels") or {}).values())
        wav = " / ".join("%.1f" % k["mean_waves_per_simd"] for k in ks if k.get("mean_waves_per_simd"))
        clk = " / ".join("%.2f" % k["held_clock_GHz"] for k in ks if k.get("held_clock_GHz") and k["held_clock_GHz"] < 2.6)
        t = c.get("polymul_ms") or c["forward_ms"]
        return "| %s | %s | %.3f (%.2f) | %s | %s | %s | **%s**: %s%s |" % (
            name, ("%.1f µs" % (t * 1e3)) if t < 0.1 else ("%.3f ms" % t), r["frac"], r["frac_ceiling"],
            ("%.3f GB (%.3f)" % (r["traffic"] / 1e9, r["traffic_ratio_to_algorithmic"])) if r.get("traffic") else "not quoted",
            ("%.2f; %.2f" % (v["instr_per_butterfly"], v["frac_at_held_clock"])) if v else "not quoted",
            ("%s; %s GHz" % (wav, clk or "n/a (launch too short for the counter quotient)")) if v else "—", r["bound"], r["bound_evidence"], extra)

    out.append(row("2: N = 2^12, 32-bit prime, batch 1024 (`%s_cfg2_*`)" % R, rows["cfg2"]))
    out.append(row("2 at a saturating batch (65536) (`%s_cfg2_sat_*`)" % R, rows["cfg2c"]))
    hr, hv = head["roofline"], head["roofline"]["valu"]
    sq = json.load(open(P("%s_sq_counters.json" % R)))
    waves = " / ".join("%.1f" % sq["kernels"][k]["mean_waves_per_simd"] for k in hv["kernels"])
    out.append("| 3: N = 2^16, Goldilocks, batch 4096 = the headline (`%s_bench.json`) | %.3f ms | %.3f (0.50) | %.3f GB (%.3f) | %.2f; %.2f | %s; %s GHz | "
               "**valu**: the table above |" % (R, head["ms_per_step"], hr["frac"], hr["traffic"] / 1e9, hr["traffic"] / hr["algorithmic_bytes_per_launch"],
                                                hv["instr_per_butterfly_mean"], hv["frac_at_held_clock"], waves,
                                                " / ".join("%.2f" % h for h in hv["held_clock_GHz"])))
    out.append(row("4: N = 2^20 negacyclic product, Goldilocks, batch 512, 9 N convention (`%s_cfg4_*`)" % R, rows["cfg4"]))
    return "\n".join(out) + "\n"


def replace(s, name, body):
    a, b = "<!-- generated:%s -->" % name, "<!-- /generated:%s -->" % name
    i, j = s.index(a) + len(a), s.index(b)
    return s[:i] + "\n" + body + s[j:]


def main():
    spread = sys.argv[1] if len(sys.argv) > 1 else "2.37–2.51 M across the boxes of the collections so far: the clock each chip holds under the cap"
    path = os.path.join(ROOT, "DESIGN.md")
    s = open(path).read()
    h, c = headline(spread), configs()
    s = replace(s, "headline", h)
    s = replace(s, "configs", c)
    m64 = jl(P("%s_bench_m64.jsonl" % R))
    gl = [x for x in m64 if x["logn"] == 16 and "goldi" in x["field"]][0]
    p62 = [x for x in m64 if x["logn"] == 16 and "62-bit" in x["field"]][0]
    s = re.sub(r"N = 2\^16, batch 4096: \*\*[\d.]+ M NTT/s = [\d.]+ × Goldilocks in the same run \([\d.]+ M\)\*\*",
               "N = 2^16, batch 4096: **%.2f M NTT/s = %.2f × Goldilocks in the same run (%.2f M)**" % (
                   p62["fwd_NTT_per_s"] / 1e6, p62["fwd_NTT_per_s"] / gl["fwd_NTT_per_s"], gl["fwd_NTT_per_s"] / 1e6), s)
    open(path, "w").write(s)
    print(h)
    print(c)


if __name__ == "__main__":
    main()
