#!/usr/bin/env python3
"""Instruction mix of a pass kernel's batch loop, from hipcc's device assembly (no GPU needed).

usage: isa_loop_stats.py file.s 'mangled-name-substring' [...]
       (make the .s with: hipcc -std=c++17 -O3 --offload-arch=gfx950 -S --cuda-device-only -Intt_aie_amd/csrc kernels_gl_fwd.hip)

The batch loop is taken to be the innermost backward branch with the most instructions between its target label and
the branch.  Prints the counts per class (VALU / v_mad_u64 / cheap VALU / SALU / LDS / VMEM / waits) per iteration."""
import collections
import re
import sys

CHEAP = ("v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_xor_b32", "v_or_b32", "v_not_b32")


def func(lines, sub):
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sub in l.split(":")[0])
    end = start
    while not lines[end].strip().startswith("s_endpgm"):
        end += 1
    return lines[start:end + 1]


def classify(op):
    if op.startswith("v_mad_u64"):
        return "valu_mad64"
    if op.startswith("v_"):
        return "valu_cheap" if op.startswith(CHEAP) and not op.endswith("_co_u32") else "valu_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    lines = open(sys.argv[1]).read().split("\n")
    for sub in sys.argv[2:]:
        f = func(lines, sub)
        labels = {l.split(":")[0]: i for i, l in enumerate(f) if re.match(r"^\.LBB\d+_\d+:", l)}
        loops = []
        for i, l in enumerate(f):
            m = re.search(r"s_cbranch\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
            if m:
                t = labels.get(m.group(1) or m.group(2))
                if t is not None and t < i:
                    loops.append((i - t, t, i))
        loops.sort(reverse=True)
        print(sub)
        for size, t, i in loops[:3]:
            cnt = collections.Counter()
            ops = collections.Counter()
            for l in f[t:i + 1]:
                l = l.strip()
                if not l or l.startswith((";", ".", "//")) or l.endswith(":"):
                    continue
                op = l.split()[0]
                cnt[classify(op)] += 1
                ops[op] += 1
            valu = cnt["valu_mad64"] + cnt["valu_cheap"] + cnt["valu_other"]
            # wall cost model of profiles/r01_microbench2_valu_forms.txt: 2.2 / 1.1 / 1.75 ns per wave-instruction per SIMD
            ns = 2.2 * cnt["valu_mad64"] + 1.1 * cnt["valu_cheap"] + 1.75 * cnt["valu_other"]
            print("  loop lines %d-%d: VALU %d (mad64 %d, cheap %d, other %d; model %.0f ns/iter)  %s" % (
                t, i, valu, cnt["valu_mad64"], cnt["valu_cheap"], cnt["valu_other"], ns,
                " ".join("%s %d" % kv for kv in sorted(cnt.items()) if not kv[0].startswith("valu"))))
            print("    top ops:", ", ".join("%s %d" % kv for kv in ops.most_common(14)))


if __name__ == "__main__":
    main()
