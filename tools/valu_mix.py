#!/usr/bin/env python3
"""Instruction mix, by issue-cost class, of the generated butterfly streams (no GPU needed).

usage: valu_mix.py > profiles/rNN_valu_mix.json

The streams of csrc/gl_asm.h are emitted by tools/gen_gl_asm.py from explicit instruction lists, so the mix per butterfly
is exact: this tool asks the generator for the list of one butterfly of every kind and sorts the VALU opcodes into the classes
tools/valu_issue_cost.hip measures.  bench.py (valu_roofline) prices `instr_per_butterfly` of a pass kernel as
    stream mix x measured cycles per class  +  (SQ_INSTS_VALU per butterfly - stream length) x the `overhead` price
(the instructions around the streams are addressing and register moves).  Stamped with the kernel-source hash like the
counter summaries: bench.py quotes it only for the tree it was made from.

Reference analogue: profile/plot_efficiency.py:44-46 prices every operation of 5.5*N*log2(N) the same; this is the same
idea with measured per-class costs."""
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

CARRY = ("v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_subbrev_co_u32", "v_subrev_co_u32")
PLAIN = ("v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32")


def classify(text: str, twiddle_in_sgpr: bool = False) -> str | None:
    """Issue-cost class of one instruction text of a stream (None for SALU)."""
    op = text.split()[0]
    if op.startswith("s_"):
        return None
    if op == "v_mad_u64_u32":
        # the column pass holds its twiddles in SGPRs: the products whose multiplicand is a twiddle half read one
        return "mad64_s" if twiddle_in_sgpr and ("%[t0_" in text or "%[t1_" in text) else "mad64"
    if op in CARRY:
        return "carry"
    if op.startswith("v_cmp") and op.endswith("_u64"):
        return "cmp64"
    if op == "v_cndmask_b32":
        return "cndmask"
    if op in PLAIN:
        return "plain"
    return "other"


def stream_mix(kind: str, twiddle_in_sgpr: bool = False) -> dict:
    import gen_gl_asm as G

    if "32:" in kind:  # 4-byte words: kind "fwd32:any" | "fwd32:small" | "fwd32:lazy" (and inv32 / mul32)
        k, mode = kind.split(":")
        ins = G.butterfly32(k, 0, mode, G.M32_VBASE)
    else:
        ins = G.butterfly(kind, 0) if kind in ("fwd", "inv", "mul", "invs") else G.butterfly64(kind, 0)
    mix = collections.Counter()
    salu = 0
    for i in ins:
        c = classify(i.text, twiddle_in_sgpr)
        if c is None:
            salu += 1
        else:
            mix[c] += 1
    return {"valu": sum(mix.values()), "salu": salu, "mix": dict(sorted(mix.items()))}


def weighted_cycles(mix: dict, costs: dict, default: float = 4.0) -> float:
    """sum over classes of count x cycles (a class the cost table lacks is priced at `default`)."""
    return sum(n * costs.get(c, default) for c, n in mix.items())


def class_costs(issue_cost: dict, waves_per_simd: int = 4) -> dict:
    """{class: cycles per wave-instruction per SIMD} from tools/valu_issue_cost's JSON: the mean over the measured forms of a
    class at `waves_per_simd` (the pass kernels run 3.7 waves per SIMD: profiles/r03_sq_counters.json)."""
    acc = collections.defaultdict(list)
    for form in issue_cost["forms"].values():
        if "+" in form["class"]:
            continue  # mixed VALU + SALU probes are checks, not prices
        acc[form["class"]].append(form["cycles"][str(waves_per_simd)])
    return {c: sum(v) / len(v) for c, v in acc.items()}


def main():
    from ntt_aie_amd._lib import kernel_source_hash

    out = {"src_hash": kernel_source_hash(),
           "note": "VALU instructions of ONE butterfly of each generated stream (tools/gen_gl_asm.py lists), by issue-cost class "
                   "(tools/valu_issue_cost.hip measures the classes); salu = s_andn2 / s_or mask ops, issued by the scalar unit",
           "streams": {"gl_fwd_v": stream_mix("fwd"), "gl_fwd_s": stream_mix("fwd", True),
                       "gl_inv_v": stream_mix("inv"), "gl_inv_s": stream_mix("inv", True),
                       "gl_invs_v": stream_mix("invs"), "gl_mul_v": stream_mix("mul")}}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
