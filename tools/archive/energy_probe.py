#!/usr/bin/env python3
"""Board power while ONE vector instruction form saturates every SIMD (tools/microbench2.hip, power mode), sampled with
rocm-smi: dynamic power and energy per wave-instruction by instruction class.  Supports DESIGN.md section 3.4: under the
1400 W cap an instruction costs what it draws, not only the cycles it issues for.
usage (GPU box): python3 tools/energy_probe.py [seconds per kernel]"""
import os, re, subprocess, sys, threading, time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SEC = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
exe = "/tmp/microbench2_power"
subprocess.check_call(["hipcc", "-O2", "--offload-arch=gfx950", os.path.join(ROOT, "tools", "microbench2.hip"), "-o", exe])


def power():
    out = subprocess.run(["rocm-smi", "-d", "0", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
    pw = re.search(r"Graphics Package Power \(W\): ([\d.]+)", out)
    sc = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
    return (float(pw.group(1)) if pw else None, int(sc.group(1)) if sc else None)


idle = [power()[0] for _ in range(5)]
idle = sum(idle) / len(idle)
print("idle %.0f W" % idle, flush=True)
print("%-22s %8s %8s %8s %10s %12s" % ("instruction", "ns/instr", "W", "sclk", "W dynamic", "nJ/wave-instr"))
for k in ("k_mov_b32", "k_add_u32", "k_and_b32", "k_add_sgpr", "k_min_u32", "k_add_co_sgpr", "k_addc_vcc", "k_subbrev_sgpr", "k_cndmask_sgpr",
          "k_cmp_le_u64_sconst", "k_cmp_ne_u32_sgpr", "k_mul_lo_u32", "k_mul_hi_u32", "k_mad_u64_u32_c0", "k_mad_u64_u32", "k_lshl_add_u64",
          "k_fma_f32", "k_pk_fma_f32", "k_fma_f64"):
    samples = []
    stop = [False]

    def sampler():
        time.sleep(0.8)  # ramp
        while not stop[0]:
            samples.append(power())
            time.sleep(0.15)

    t = threading.Thread(target=sampler)
    t.start()
    out = subprocess.run([exe, "power", k, str(SEC)], capture_output=True, text=True).stdout
    stop[0] = True
    t.join()
    ns = float(out.split()[1])
    pw = sorted(s[0] for s in samples if s[0])[len(samples) // 4:]  # drop the ramp's low samples
    w = sum(pw) / len(pw)
    sc = sorted(s[1] for s in samples if s[1])
    dyn = w - idle
    print("%-22s %8.3f %8.0f %8d %10.0f %12.3f" % (k[2:], ns, w, sc[len(sc) // 2], dyn, dyn * ns / 1024.0), flush=True)
