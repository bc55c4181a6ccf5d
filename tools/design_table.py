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
    sa = v.get("statement_alone_steady_state") or {}
    if not sa:  # the line was printed before the round's stream-occupancy table was in profiles/: the same figures from the file itself
        st = bench.statement_steady_state()
        rows = {str(k): x for k, x in st["cycles_per_butterfly_by_waves_per_simd"].items()}
        best = min(x for k, x in rows.items() if int(k) >= 4)
        sa = {"cycles_per_butterfly_by_waves_per_simd": rows, "cycles_per_butterfly_at_4_or_more_waves": best,
              "kernel_over_statement": [k / best for k in v["kernel_cycles_per_wave_butterfly_per_simd"]]}
    # a 72-character label, then: ns, clock GHz, cycles[, TFLOP/s]
    vpk = {l.split()[0]: float(l[72:].split()[2]) for l in open(P("%s_valu_peak.txt" % R)) if l[:2] == "v_"}
    rk = d["ranks"][0]
    ps = json.load(open(P("%s_phase_stamps.json" % R)))
    sp = [p["split"] for p in ps["passes"]]

    def share(i, key):
        return 100 * next(v_["share"] for k_, v_ in sp[i].items() if k_.startswith(key))

    return """| quantity (`profiles/{R}_bench.json`, `{R}_kernel_stats.csv`, `{R}_pmc_traffic.json`, `{R}_sq_counters.json`, `{R}_phase_stamps.json`) | value |
|---|---|
| throughput | **{val:.2f} M NTT/s = {bf:.2f}e12 butterflies/s, {ms:.3f} ms per step** ({spread}) |
| verification in the same run | rank 0 on `{bus}`: inverse(forward(x)) == x over the whole shard, out[b][0] == Σ a[b][:] mod p on {rows} rows: `all_ranks_verified` {ver} |
| pass kernels, hipEvents / rocprofv3 | CONTIG {p0:.3f} / {k0:.3f} ms, column {p1:.3f} / {k1:.3f} ms |
| `roofline.achieved / peak / frac` (contract: algorithmic bytes over 8 TB/s) | {ach:.2f} TB/s / 8 TB/s = **{frac:.3f}**; `frac_step` (over the line's own `ms_per_step`, not clamped) {fs:.3f}; `frac_ceiling` 0.5 (two passes) |
| `frac_of_practical_hbm` | two trips at the same-run device-copy rate ({cp:.2f} TB/s) would take {pf:.3f} ms: the step is at **{fp:.2f}** of that |
| `roofline.traffic` (PMC, forward kernels) | {tr:.3f} GB = {trr:.2f} × algorithmic: two trips, no over-fetch within a pass |
| each pass's stream rate | {s0:.2f} / {s1:.2f} TB/s = {f0:.2f}–{f1:.2f} of peak = {c0:.2f}–{c1:.2f} of the same-process device copy |
| the vector ALU's **measured** throughput (`{R}_valu_peak.txt`, `tools/valu_peak.hip`: launch duration × in-kernel clock ÷ wave-instructions per SIMD, 12 generations of workgroups) | `v_add_co_u32` {vadd:.2f}, `v_addc_co_u32` {vaddc:.2f}, `v_mad_u64_u32` {vmad:.2f}, `v_mul_lo_u32` {vmul:.2f}, `v_cndmask_b32` by SGPR pair {vcnd:.2f} cycles per wave64 instruction; `v_mov_b32` {vmov:.2f}, `v_add_u32` {vaddu:.2f}; sanity: `v_pk_fma_f32` {vpk:.2f} cycles = the data sheet's 157 TFLOP/s at 2.4 GHz; `v_mul_f32` {vmulf:.2f} (the guide's two-operand rate), every FMA form {vfma:.2f}–{vfmac:.2f} (round 6: three operand reads cost that whatever the encoding). Nominal prices (`tools/hw.py`): **4 cycles** for the statements' VOP3-class forms, **2** for plain moves / adds |
| `roofline.valu` on that throughput | {ipb:.2f} VALU per butterfly ({i0:.2f} / {i1:.2f}, 2 of them moves) = {pc0:.1f} / {pc1:.1f} cycles at peak; the kernels take {kc0:.1f} / {kc1:.1f} cycles per wave-butterfly per SIMD ({h0:.2f}–{h1:.2f} GHz under the counter run, {w0:.1f} / {w1:.1f} waves per SIMD): `frac_of_peak_at_held_clock` **{fh0:.2f} / {fh1:.2f}**; peak {pk:.2f}e12 bf/s at 2.4 GHz, {f24:.2f} of it in wall-clock terms |
| … and the butterfly statement ALONE in steady state (`{R}_stream_occupancy.txt`, last table) | {st1:.0f} / {st2:.0f} / {st3:.0f} / {st4:.0f} / {st8:.0f} cycles per butterfly per SIMD at 1 / 2 / 3 / 4 / 8 resident waves: from four waves on it runs AT the unit's throughput ({stb:.0f} against 84 at nominal prices); the kernels take {ks0:.2f} / {ks1:.2f} × that |
| **`roofline.bound` = `{bound}`** (decided by `bench.decide_bound` from these numbers) | {bd} |
| what holds the clock | the 1400 W board cap: transform {pt:.0f} W at {ct:.2f} GHz; its VALU work alone (loads from L2, no stores) {pa:.0f} W at {ca:.2f} GHz; a copy of its bytes alone {pc:.0f} W (`profiles/{R}_power_probe.txt`) |
| VALU floor (same kernels, loads from L2, no stores; measured in the bench run) | {fl0:.3f} + {fl1:.3f} ms (cycle view: `{R}_sq_real_vs_floor.txt`) |
| where a wave's cycles go (s_memtime stamps at every phase boundary, diagnostic build; stamping costs +{so0:.0f} % / +{so1:.0f} %) | CONTIG: butterfly rounds {a0:.0f} %, LDS exchanges {b0:.0f} %, tile wait + prefetch issue + stores {c0_:.0f} %, sync + loop {d0:.0f} % of a steady-state iteration ({it0:.0f} cycles for 32 wave-butterflies); column: {a1:.0f} % / {b1:.0f} % / {c1_:.0f} % (its loads are not prefetched) / {d1:.0f} % ({it1:.0f} cycles for 64) |
| CPU baseline (oracle port, same run, **the GPU's own input rows, never beyond the batch**) | {c1t:.0f} NTT/s on 1 thread; {call:.2f} k on the {cores} cores the box's cgroup quota allows (affinity mask {aff}); {crows} rows |
| inverse (config 3's second leg) | {inv:.2f} ms, {invr:.3f} × forward, round trip identical |
""".format(R=R, val=d["value"] / 1e6, bf=d["butterflies_per_s"] / 1e12, ms=d["ms_per_step"], spread=spread,
           bus=rk.get("pci_bus_id") or rk.get("uuid"), rows=rk["rows_sampled"], ver=str(d["all_ranks_verified"]).lower(),
           p0=r["pass_ms"][0], k0=k0, p1=r["pass_ms"][1], k1=k1, ach=r["achieved"] / 1e3, frac=r["frac"], fs=r["frac_step"],
           pf=r["practical_hbm_floor_ms"], fp=r["frac_of_practical_hbm"],
           tr=r["traffic"] / 1e9, trr=r["traffic"] / r["algorithmic_bytes_per_launch"],
           s0=r["pass_stream_GBs"][0] / 1e3, s1=r["pass_stream_GBs"][1] / 1e3, f0=min(r["pass_stream_frac"]), f1=max(r["pass_stream_frac"]),
           c0=min(r["pass_stream_frac_of_device_copy"]), c1=max(r["pass_stream_frac_of_device_copy"]), cp=r["device_copy"]["GBs"] / 1e3,
           ipb=v["instr_per_butterfly_mean"], i0=v["instr_per_butterfly"][0], i1=v["instr_per_butterfly"][1],
           pk=v["peak_butterflies_per_s"] / 1e12, f24=v["frac_of_peak_at_2.4GHz"],
           fh0=v["frac_of_peak_at_held_clock_per_pass"][0], fh1=v["frac_of_peak_at_held_clock_per_pass"][1],
           kc0=v["kernel_cycles_per_wave_butterfly_per_simd"][0], kc1=v["kernel_cycles_per_wave_butterfly_per_simd"][1],
           h0=min(v["held_clock_GHz"]), h1=max(v["held_clock_GHz"]), w0=v["mean_waves_per_simd"][0], w1=v["mean_waves_per_simd"][1],
           vadd=vpk["v_add_co_u32"], vaddc=vpk["v_addc_co_u32"], vmad=vpk["v_mad_u64_u32"], vmul=vpk["v_mul_lo_u32"], vcnd=vpk["v_cndmask_b32"],
           vmov=vpk["v_mov_b32"], vaddu=vpk["v_add_u32"], vpk=vpk["v_pk_fma_f32"], vmulf=vpk["v_mul_f32"],
           vfma=min(vpk["v_fma_f32"], vpk["v_fma_f32_2src"], vpk["v_fmac_f32"]), vfmac=max(vpk["v_fma_f32"], vpk["v_fma_f32_2src"], vpk["v_fmac_f32"]),
           pc0=v["peak_cycles_per_butterfly"][0], pc1=v["peak_cycles_per_butterfly"][1],
           st1=sa["cycles_per_butterfly_by_waves_per_simd"]["1"], st2=sa["cycles_per_butterfly_by_waves_per_simd"]["2"],
           st3=sa["cycles_per_butterfly_by_waves_per_simd"]["3"], st4=sa["cycles_per_butterfly_by_waves_per_simd"]["4"],
           st8=sa["cycles_per_butterfly_by_waves_per_simd"]["8"], stb=sa["cycles_per_butterfly_at_4_or_more_waves"],
           ks0=sa["kernel_over_statement"][0], ks1=sa["kernel_over_statement"][1], bound=r["bound"], bd=r["bound_detail"],
           pt=pt, ct=ct, pa=pa, ca=ca, pc=pc, fl0=r["valu_floor_pass_ms"][0], fl1=r["valu_floor_pass_ms"][1],
           so0=100 * (ps["overhead"]["stamped_over_product"][0] - 1), so1=100 * (ps["overhead"]["stamped_over_product"][1] - 1),
           a0=share(0, "compute"), b0=share(0, "exchange"), c0_=share(0, "memory"), d0=share(0, "sync"), it0=ps["passes"][0]["iteration_cycles_mean_steady"],
           a1=share(1, "compute"), b1=share(1, "exchange"), c1_=share(1, "memory"), d1=share(1, "sync"), it1=ps["passes"][1]["iteration_cycles_mean_steady"],
           c1t=cb["value_1thread"], call=cb["value"] / 1e3, cores=cb["cores"], aff=cb["host_affinity_cores"], crows=cb.get("sample_rows", 0),
           inv=d["inverse"]["ms_per_step_median"], invr=d["inverse"]["vs_forward_median"])


def configs():
    """configs 2 / 3 / 4: one row each, the same columns.  Configs 2 and 4 are the entries of the SAME driver-run line as the headline
    (bench.py's `configs` key); config 2 at a saturating batch comes from tools/bench_configs.py."""
    rows = {c["config"].split(":")[0]: c for c in jl(P("%s_bench_all_configs.jsonl" % R))}
    head = jl(P("%s_bench.json" % R))[-1]
    line = {e["key"]: e for e in head["configs"]}
    out = ["| config | time per operation | `frac` of 8 TB/s (ceiling) | HBM bytes, PMC (÷ algorithmic) | VALU per butterfly; at the unit's measured throughput ÷ kernel cycles | "
           "waves / SIMD; held clock | `bound` | verified in the line |", "|---|---|---|---|---|---|---|---|"]

    def row(name, t, r, verified):
        v = r.get("valu") or {}
        wav = " / ".join("%.1f" % w for w in (v.get("mean_waves_per_simd") or []) if w)
        clk = " / ".join("%.2f" % h for h in (v.get("held_clock_GHz") or []) if h)  # (impossible quotients are nulled at the source: bench.sane_clocks)
        return "| %s | %s | %.3f (%.2f) | %s | %s | %s | **%s**: %s | %s |" % (
            name, ("%.1f µs" % (t * 1e3)) if t < 0.1 else ("%.3f ms" % t), r["frac"], r["frac_ceiling"],
            ("%.3f GB (%.3f)" % (r["traffic"] / 1e9, r["traffic_ratio_to_algorithmic"])) if r.get("traffic") else "not quoted",
            ("%.2f; %s" % (v["instr_per_butterfly"], ("%.2f" % v["frac_of_peak_at_held_clock"]) if v.get("frac_of_peak_at_held_clock") else "n/a")) if v else "not quoted",
            ("%s; %s GHz" % (wav, clk or "n/a (launch too short for the counter quotient)")) if v else "—", r["bound"], r["bound_detail"], verified)

    def ver(e):
        return "; ".join("%s %s" % (k, str(v_).lower()) for k, v_ in e["verification"].items() if isinstance(v_, bool))

    e2, e4 = line["cfg2"], line["cfg4"]
    out.append(row("2: N = 2^12, 32-bit prime, batch 1024 (`%s_bench.json: configs[0]`, `%s_cfg2_*`)" % (R, R), e2["ms"], e2["roofline"], ver(e2)))
    c2s = rows["cfg2c"]
    out.append(row("2 at a saturating batch (65536) (`%s_bench_all_configs.jsonl`, `%s_cfg2_sat_*`)" % (R, R), c2s["forward_ms"], c2s["roofline"], "tools/bench_configs.py (not in the line)"))
    hr, hv = head["roofline"], head["roofline"]["valu"]
    out.append("| 3: N = 2^16, Goldilocks, batch 4096 = the headline (`%s_bench.json`) | %.3f ms | %.3f (0.50) | %.3f GB (%.3f) | %.2f; %.2f | %s; %s GHz | "
               "**%s**: the table above | round trip + coefficient sum, every rank |" % (
                   R, head["ms_per_step"], hr["frac"], hr["traffic"] / 1e9, hr["traffic"] / hr["algorithmic_bytes_per_launch"],
                   hv["instr_per_butterfly_mean"], hv["frac_of_peak_at_held_clock"], " / ".join("%.1f" % w for w in hv["mean_waves_per_simd"]),
                   " / ".join("%.2f" % h for h in hv["held_clock_GHz"]), hr["bound"]))
    out.append(row("4: N = 2^20 negacyclic product, Goldilocks, batch 512, 9 N convention (`%s_bench.json: configs[1]`, `%s_cfg4_*`)" % (R, R), e4["ms"], e4["roofline"], ver(e4)))
    if "cfg5_shard" in line:
        e5 = line["cfg5_shard"]
        out.append(row("5's per-GPU shard: N = 2^16, Goldilocks, 8192 of the 8-GPU job's 65536 rows on ONE GPU (`%s_bench.json: configs[2]`; the 8-rank job itself has "
                       "never run: no node)" % R, e5["ms"], e5["roofline"], ver(e5)))
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
