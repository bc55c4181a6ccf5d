#!/usr/bin/env python3
"""From a chrome-trace file of one compute unit's waves (tools/phase_stamps.py --trace-prefix): per SIMD, the share of time with k waves
inside a butterfly round, and the SIMD-cycles one butterfly takes while exactly k waves are in a round -- to be read beside the bare
statement's figures (profiles/r05_stream_occupancy.txt: 141.7 / 106.2 / 59.0 / 72.2 cycles per butterfly at 1 .. 4 computing waves).
usage: trace_occupancy.py profiles/trace/trace_mi355x_n16_contig.json [butterflies per round ...]   (default 12 12 8: rounds 3+3+2 on 8 words)"""
import collections
import json
import sys


def analyse(events, bf_per_round):
    rows = {e["tid"]: e["args"]["name"] for e in events if e["ph"] == "M" and e["name"] == "thread_name"}
    simd_of = {tid: int(n.split()[1]) for tid, n in rows.items()}
    bf = {"round %d butterflies" % r: b for r, b in enumerate(bf_per_round)}
    open_, iv = {}, []
    for e in events:
        if e["ph"] == "B":
            open_[(e["tid"], e["name"])] = e["ts"]
        elif e["ph"] == "E":
            iv.append((e["tid"], e["name"], open_.pop((e["tid"], e["name"])), e["ts"]))
    out = {}
    for simd in sorted(set(simd_of.values())):
        comp = [(a, b, bf[n]) for t, n, a, b in iv if simd_of[t] == simd and n in bf and b > a]
        ev = sorted([(a, 1) for a, _, _ in comp] + [(b, -1) for _, b, _ in comp])
        k, last, kdur, pts = 0, None, collections.Counter(), []
        for t, dk in ev:
            if last is not None and t > last:
                kdur[k] += t - last
                pts.append((last, t, k))
            k += dk
            last = t
        work = collections.Counter()
        for a, b, n in comp:  # a round's butterflies spread evenly over its interval
            r = n / (b - a)
            for x, y, kk in pts:
                lo, hi = max(a, x), min(b, y)
                if hi > lo:
                    work[kk] += r * (hi - lo)
        tot = float(sum(kdur.values())) or 1.0
        out[simd] = {"share_of_time_with_k_computing_waves": {k_: kdur[k_] / tot for k_ in sorted(kdur)},
                     "simd_cycles_per_butterfly_at_k": {k_: kdur[k_] / work[k_] for k_ in sorted(work) if work[k_] > 0},
                     "mean_computing_waves": sum(k_ * v for k_, v in kdur.items()) / tot}
    return out


if __name__ == "__main__":
    ev = json.load(open(sys.argv[1]))
    bfr = [int(v) for v in sys.argv[2:]] or [12, 12, 8]
    print("# %s  (butterflies per thread and round: %s)" % (sys.argv[1], bfr))
    for simd, r in analyse(ev, bfr).items():
        print("SIMD %d  mean computing waves %.2f  time share by k: %s  SIMD-cycles per butterfly at k: %s" % (
            simd, r["mean_computing_waves"], {k: round(v, 3) for k, v in r["share_of_time_with_k_computing_waves"].items()},
            {k: round(v, 1) for k, v in r["simd_cycles_per_butterfly_at_k"].items()}))
