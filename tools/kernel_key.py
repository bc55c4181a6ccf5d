"""Identity of a pass kernel in a rocprofv3 CSV: the FULL `PassCfg<...>` template argument list, direction included.

Round 2 keyed counters by (contig | col, LOG_M) only; once bench.py also launched the inverse transform the inverse kernels
overwrote / mixed into the forward kernels' keys ("What's weak" 1).  Every summary tool and bench.py now go
through this one parser, and a CPU unit test (tests/test_profile_tools.py) feeds it a CSV holding both directions.

PassCfg template arguments (csrc/pass.h): <F, LOG_M, LOG_C, CONTIG, INV, PRELOAD_MASK, LOG_E, LOG_NT, ALLOW_DMA>.
pass_kernel has a second template argument since round 3: pass_kernel<Cfg, SC>, SC = true for the scaled inverse whose N^-1 is
folded into stage 0 (a different kernel: other registers, N/2 more products).  SC is part of the key (round 4; before, the two
inverse kernels of one shape were averaged together).
"""
from __future__ import annotations

FIELDS = ("field", "log_m", "log_c", "contig", "inv", "preload_mask", "log_e", "log_nt", "allow_dma")


def parse_pass_kernel(name: str):
    """{'cfg': normalised argument list, 'field', 'log_m', 'contig', 'inv', ..., 'key', 'short'} or None when `name` is not a
    pass kernel.  `key` = the full argument list, plus " +SC" for pass_kernel<Cfg, true> (unique per instantiation);
    `short` = pass_<contig|col>_<LOG_M>_<fwd|inv>[_sc]."""
    if "pass_kernel<" not in name or "PassCfg<" not in name:
        return None
    rest = name.split("PassCfg<", 1)[1]
    args = [a.strip() for a in rest.split(">", 1)[0].split(",")]
    # what follows the first PassCfg<...>: ", true>(" / ", false>(" in a demangled kernel name; nothing in a stored key
    tail = rest.split(">", 1)[1].lstrip() if ">" in rest else ""
    sc = tail.startswith(",") and tail[1:].lstrip().startswith("true")
    sc = sc or tail.startswith("+SC")
    if len(args) < 5:
        return None
    d = dict(zip(FIELDS, args))
    out = {
        "cfg": "PassCfg<%s>" % ", ".join(args),
        "field": d["field"].split("::")[-1],
        "log_m": int(d["log_m"]),
        "log_c": int(d["log_c"]),
        "contig": d["contig"] == "true",
        "inv": d["inv"] == "true",
        "log_e": int(d["log_e"]) if "log_e" in d else None,
        "log_nt": int(d["log_nt"]) if "log_nt" in d else None,
    }
    out["sc"] = bool(sc)
    out["key"] = out["cfg"] + (" +SC" if sc else "")
    out["short"] = "pass_%s_%d_%s%s" % ("contig" if out["contig"] else "col", out["log_m"], "inv" if out["inv"] else "fwd",
                                        "_sc" if sc else "")
    return out


def parse_kernel(name: str):
    """parse_pass_kernel() for pass_kernel<...>, and the product's fused middle pass
    product_kernel<PassCfg<inverse ...>, PassCfg<forward ...>>: key "product PassCfg<...>" (the inverse leg's argument list),
    short product_<LOG_M>, kind "product".  `stage_legs` = how many LOG_M-stage networks one launch runs per polynomial
    (1 for a pass; 3 for the product: inverse of a, inverse of b, forward of the product).  None for any other kernel."""
    if "product_kernel<" in name and "PassCfg<" in name:
        args = [a.strip() for a in name.split("PassCfg<", 1)[1].split(">", 1)[0].split(",")]
        if len(args) < 5:
            return None
        d = dict(zip(FIELDS, args))
        return {"cfg": "PassCfg<%s>" % ", ".join(args), "field": d["field"].split("::")[-1], "log_m": int(d["log_m"]),
                "log_c": int(d["log_c"]), "contig": True, "inv": False, "sc": False, "kind": "product", "stage_legs": 3,
                "key": "product PassCfg<%s>" % ", ".join(args), "short": "product_%d" % int(d["log_m"])}
    pk = parse_pass_kernel(name)
    if pk is not None:
        pk["kind"], pk["stage_legs"] = "pass", 1
    return pk


def forward_entry(kernels: dict, contig: bool, log_m: int, field: str = "FieldGL"):
    """The ONE forward-direction entry of a summary's `kernels` table for this pass shape, or (None, reason).
    bench.py quotes a counter only through this function: an inverse kernel can never be returned."""
    hits = []
    for key, v in kernels.items():
        k = parse_pass_kernel("pass_kernel<" + key) if key.startswith("PassCfg<") else None
        if k is None:
            continue
        if k["inv"] or k["contig"] != contig or k["log_m"] != log_m or k["field"] != field:
            continue
        hits.append((key, v))
    if not hits:
        return None, "no forward %s kernel with %d stages (%s) in the summary" % ("CONTIG" if contig else "column", log_m, field)
    if len(hits) > 1:
        return None, "ambiguous: %d forward kernels match (%s)" % (len(hits), "; ".join(h[0] for h in hits))
    return hits[0], None
