#!/bin/bash
# NEGATIVE RESULT, kept for provenance.  Does doubling the waves per SIMD pay on a headline kernel?  A side build
# (profiles/r04_col8_experiment.patch applied to a copy of ntt_aie_amd/csrc + tools/gen_gl_asm.py -> ab/libntt_col8.so) runs the
# Goldilocks 8-stage column pass as PassCfg<FieldGL, 8, 4, false, INV, 0xF, 3, 9>: 512 threads x 8 words, radix-8 rounds 3 + 3 + 2
# (two exchanges instead of one), butterfly scratch at v[40:63], __launch_bounds__(512, 8) = 64 VGPRs (8 spilled), 40 KiB of LDS:
# four workgroups = 8 waves per SIMD where the shipped radix-16 kernel holds 4.  Validated in the host index model first (natural
# layout), then same-process A/B, outputs compared.  -> profiles/r04_ab_col8.txt: the column pass 0.818 -> 1.008 ms (+23 %).
cd "$GRAFT_REPO_ROOT"
G=18446744069414584321
echo "## whole forward transform, outputs compared"
python3 tools/ab_latency.py --logn 16 --p $G --g 7 --word-bytes 8 --batch 4096 --rounds 7 --k 10 base=ntt_aie_amd/libntt_hip.so col8=ab/libntt_col8.so 2>&1 | grep -v amdgpu.ids
python3 tools/ab_latency.py --logn 16 --p $G --g 7 --word-bytes 8 --batch 8192 --rounds 5 --k 10 base=ntt_aie_amd/libntt_hip.so col8=ab/libntt_col8.so 2>&1 | grep -v amdgpu.ids
python3 tools/ab_latency.py --logn 20 --p $G --g 7 --word-bytes 8 --batch 256 --rounds 5 --k 10 base=ntt_aie_amd/libntt_hip.so col8=ab/libntt_col8.so 2>&1 | grep -v amdgpu.ids
python3 tools/ab_latency.py --logn 16 --p $G --g 7 --word-bytes 8 --batch 64 --rounds 5 --k 20 base=ntt_aie_amd/libntt_hip.so col8=ab/libntt_col8.so 2>&1 | grep -v amdgpu.ids
echo "## per pass (hipEvents), headline"
python3 tools/ab_pass.py --rounds 7 --reps 5 base=ntt_aie_amd/libntt_hip.so col8=ab/libntt_col8.so 2>&1 | grep -v amdgpu.ids
