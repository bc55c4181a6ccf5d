// NOTE (round 5): the "cycles" this tool prints are the MEDIAN wave's own cycle count of a ONE-generation launch divided by (waves per SIMD x
// instructions) -- they presume that every wave of the launch runs side by side from start to end, which does not hold (its own "ns" column, wall
// clock, says 4.3 cycles where "cycles" says 1.99).  They are a per-wave figure (useful at ONE wave: the pipeline's issue interval), NOT a throughput:
// tools/valu_peak.hip measures the throughput (4 cycles per VOP3-class form, 2 per plain move / add) and tools/hw.py prices the rooflines on it.
//
// valu_issue_cost.hip -- issue cost, in SHADER CYCLES per wave-instruction per SIMD, of every VALU form the Goldilocks
// butterfly streams (csrc/gl_asm.h) are made of (1 .. 8 waves per SIMD), and of the streams themselves (1 .. 4: they pin v104+).
//
// Why: bench.py's VALU roofline priced every VALU instruction at 4 cycles.  The forms differ:
// gfx950's vector ALU retires plain 32-bit ops of a wave64 in 2 cycles once two waves share the SIMD, VOP3 carry forms in 4,
// v_mad_u64_u32 in more.  This program measures each form the way the kernels use it (SGPR-pair carries, SGPR or VGPR
// multiplicands, the zero-high addend pair) and the generated two-butterfly statements as a whole (a radix-8 round on 8
// register-resident words: 12 butterflies, exactly what a thread of the first pass does between two exchanges).
//
// Method: every wave stamps s_memtime (tick = shader cycle) and s_memrealtime (100 MHz) around its loop; cost =
// median-over-waves(delta cycles) / (waves per SIMD x instructions per wave).  Occupancy is FORCED: 256-thread workgroups
// (one wave per SIMD), 256 x W workgroups, each asking for floor(160 KiB / W) of LDS, so every CU holds exactly W of them.
//
// build: hipcc -O3 --offload-arch=gfx950 -I ntt_aie_amd/csrc tools/valu_issue_cost.hip -o tools/valu_issue_cost
// run:   tools/valu_issue_cost > profiles/r04_valu_issue_cost.json      (one JSON object on stdout; progress on stderr)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <string>
#include <vector>

#include "gl_asm.h"

#define REP8(x) x x x x x x x x
constexpr int ITERS = 1500;
constexpr int CUS = 256;

struct Stamp {
    unsigned long long cycles, ticks;
};

#define STAMP_BEGIN                                             \
    __builtin_amdgcn_sched_barrier(0);                          \
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(); \
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(); \
    __builtin_amdgcn_s_waitcnt(0xC07F);                         \
    __builtin_amdgcn_sched_barrier(0);
#define STAMP_END                                               \
    __builtin_amdgcn_sched_barrier(0);                          \
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(); \
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime(); \
    __builtin_amdgcn_s_waitcnt(0xC07F);                         \
    __builtin_amdgcn_sched_barrier(0);                          \
    if ((threadIdx.x & 63) == 0) st[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = Stamp{c1 - c0, r1 - r0};

// One instruction form, 8 independent register sets, 64 instructions per loop iteration.
// 32-bit regs w0..w7 = %0..%7, 64-bit regs r0..r7 = %8..%15, a = %16, b = %17 (VGPR), sg = %18 (SGPR), sp = %19 (SGPR pair)
#define KERNEL(NAME, BODY)                                                                                            \
    __global__ void __launch_bounds__(256) NAME(uint32_t *out, Stamp *st, int iters, uint32_t seed) {                 \
        uint32_t a = threadIdx.x * 2654435761u + seed, b = threadIdx.x * 40503u + 7 * seed;                           \
        uint64_t r0_ = a, r1_ = b, r2_ = a ^ b, r3_ = a + b, r4_ = 5 + a, r5_ = 6 + b, r6_ = 7 * a, r7_ = 8 * b;      \
        uint32_t w0 = a, w1 = b, w2 = a ^ b, w3 = a + b, w4 = 1 + a, w5 = 2 + b, w6 = 3 * a, w7 = 4 * b;              \
        const uint32_t sg = seed * 77u + 1;                                                                           \
        const uint64_t sp = 0xFFFFFFFF00000001ull;                                                                    \
        STAMP_BEGIN                                                                                                   \
        for (int i = 0; i < iters; ++i) {                                                                             \
            REP8(asm volatile(BODY                                                                                    \
                              : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7),       \
                                "+v"(r0_), "+v"(r1_), "+v"(r2_), "+v"(r3_), "+v"(r4_), "+v"(r5_), "+v"(r6_), "+v"(r7_) \
                              : "v"(a), "v"(b), "s"(sg), "s"(sp)                                                      \
                              : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)               \
        }                                                                                                             \
        STAMP_END                                                                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] =                                                                  \
            w0 ^ w1 ^ w2 ^ w3 ^ w4 ^ w5 ^ w6 ^ w7 ^ (uint32_t) (r0_ ^ r1_ ^ r2_ ^ r3_ ^ r4_ ^ r5_ ^ r6_ ^ r7_);        \
    }

// plain full-rate forms
KERNEL(k_mov_b32, "v_mov_b32 %0, %16\n v_mov_b32 %1, %17\n v_mov_b32 %2, %16\n v_mov_b32 %3, %17\n v_mov_b32 %4, %16\n v_mov_b32 %5, %17\n v_mov_b32 %6, %16\n v_mov_b32 %7, %17\n")
KERNEL(k_add_u32, "v_add_u32 %0, %0, %16\n v_add_u32 %1, %1, %16\n v_add_u32 %2, %2, %16\n v_add_u32 %3, %3, %16\n v_add_u32 %4, %4, %16\n v_add_u32 %5, %5, %16\n v_add_u32 %6, %6, %16\n v_add_u32 %7, %7, %16\n")
// carry forms exactly as in the streams: VOP3, carry-out / carry-in in an SGPR pair
KERNEL(k_add_co_sgpr, "v_add_co_u32 %0, s[20:21], %0, %16\n v_add_co_u32 %1, s[22:23], %1, %16\n v_add_co_u32 %2, s[24:25], %2, %16\n v_add_co_u32 %3, s[26:27], %3, %16\n v_add_co_u32 %4, s[20:21], %4, %16\n v_add_co_u32 %5, s[22:23], %5, %16\n v_add_co_u32 %6, s[24:25], %6, %16\n v_add_co_u32 %7, s[26:27], %7, %16\n")
KERNEL(k_addc_co_sgpr, "v_addc_co_u32 %0, s[20:21], %0, %16, s[20:21]\n v_addc_co_u32 %1, s[22:23], %1, %16, s[22:23]\n v_addc_co_u32 %2, s[24:25], %2, %16, s[24:25]\n v_addc_co_u32 %3, s[26:27], %3, %16, s[26:27]\n v_addc_co_u32 %4, s[20:21], %4, %16, s[20:21]\n v_addc_co_u32 %5, s[22:23], %5, %16, s[22:23]\n v_addc_co_u32 %6, s[24:25], %6, %16, s[24:25]\n v_addc_co_u32 %7, s[26:27], %7, %16, s[26:27]\n")
KERNEL(k_addc_co_zero, "v_addc_co_u32 %0, s[20:21], %0, 0, s[20:21]\n v_addc_co_u32 %1, s[22:23], %1, 0, s[22:23]\n v_addc_co_u32 %2, s[24:25], %2, 0, s[24:25]\n v_addc_co_u32 %3, s[26:27], %3, 0, s[26:27]\n v_addc_co_u32 %4, s[20:21], %4, 0, s[20:21]\n v_addc_co_u32 %5, s[22:23], %5, 0, s[22:23]\n v_addc_co_u32 %6, s[24:25], %6, 0, s[24:25]\n v_addc_co_u32 %7, s[26:27], %7, 0, s[26:27]\n")
KERNEL(k_sub_co_sgpr, "v_sub_co_u32 %0, s[20:21], %0, %16\n v_sub_co_u32 %1, s[22:23], %1, %16\n v_sub_co_u32 %2, s[24:25], %2, %16\n v_sub_co_u32 %3, s[26:27], %3, %16\n v_sub_co_u32 %4, s[20:21], %4, %16\n v_sub_co_u32 %5, s[22:23], %5, %16\n v_sub_co_u32 %6, s[24:25], %6, %16\n v_sub_co_u32 %7, s[26:27], %7, %16\n")
KERNEL(k_subb_co_sgpr, "v_subb_co_u32 %0, s[20:21], %0, %16, s[20:21]\n v_subb_co_u32 %1, s[22:23], %1, %16, s[22:23]\n v_subb_co_u32 %2, s[24:25], %2, %16, s[24:25]\n v_subb_co_u32 %3, s[26:27], %3, %16, s[26:27]\n v_subb_co_u32 %4, s[20:21], %4, %16, s[20:21]\n v_subb_co_u32 %5, s[22:23], %5, %16, s[22:23]\n v_subb_co_u32 %6, s[24:25], %6, %16, s[24:25]\n v_subb_co_u32 %7, s[26:27], %7, %16, s[26:27]\n")
KERNEL(k_subbrev_co_zero, "v_subbrev_co_u32 %0, s[20:21], 0, %0, s[20:21]\n v_subbrev_co_u32 %1, s[22:23], 0, %1, s[22:23]\n v_subbrev_co_u32 %2, s[24:25], 0, %2, s[24:25]\n v_subbrev_co_u32 %3, s[26:27], 0, %3, s[26:27]\n v_subbrev_co_u32 %4, s[20:21], 0, %4, s[20:21]\n v_subbrev_co_u32 %5, s[22:23], 0, %5, s[22:23]\n v_subbrev_co_u32 %6, s[24:25], 0, %6, s[24:25]\n v_subbrev_co_u32 %7, s[26:27], 0, %7, s[26:27]\n")
// 64-bit compare against the modulus held in an SGPR pair, result in an SGPR pair
KERNEL(k_cmp_le_u64_sgpr, "v_cmp_le_u64 s[20:21], %19, %8\n v_cmp_le_u64 s[22:23], %19, %9\n v_cmp_le_u64 s[24:25], %19, %10\n v_cmp_le_u64 s[26:27], %19, %11\n v_cmp_le_u64 s[20:21], %19, %12\n v_cmp_le_u64 s[22:23], %19, %13\n v_cmp_le_u64 s[24:25], %19, %14\n v_cmp_le_u64 s[26:27], %19, %15\n")
// 0 / 1 from an SGPR-pair mask (the carry of the middle sum of the 128-bit product)
KERNEL(k_cndmask_01_sgpr, "v_cndmask_b32 %0, 0, 1, s[20:21]\n v_cndmask_b32 %1, 0, 1, s[22:23]\n v_cndmask_b32 %2, 0, 1, s[24:25]\n v_cndmask_b32 %3, 0, 1, s[26:27]\n v_cndmask_b32 %4, 0, 1, s[20:21]\n v_cndmask_b32 %5, 0, 1, s[22:23]\n v_cndmask_b32 %6, 0, 1, s[24:25]\n v_cndmask_b32 %7, 0, 1, s[26:27]\n")
// the four multiply-add forms of the 128-bit product: x*t + 0, + a register pair; VGPR and SGPR multiplicand
KERNEL(k_mad64_vv_0, "v_mad_u64_u32 %8, s[20:21], %16, %0, 0\n v_mad_u64_u32 %9, s[22:23], %16, %1, 0\n v_mad_u64_u32 %10, s[24:25], %16, %2, 0\n v_mad_u64_u32 %11, s[26:27], %16, %3, 0\n v_mad_u64_u32 %12, s[20:21], %16, %4, 0\n v_mad_u64_u32 %13, s[22:23], %16, %5, 0\n v_mad_u64_u32 %14, s[24:25], %16, %6, 0\n v_mad_u64_u32 %15, s[26:27], %16, %7, 0\n")
KERNEL(k_mad64_vv_acc, "v_mad_u64_u32 %8, s[20:21], %16, %17, %8\n v_mad_u64_u32 %9, s[22:23], %16, %17, %9\n v_mad_u64_u32 %10, s[24:25], %16, %17, %10\n v_mad_u64_u32 %11, s[26:27], %16, %17, %11\n v_mad_u64_u32 %12, s[20:21], %16, %17, %12\n v_mad_u64_u32 %13, s[22:23], %16, %17, %13\n v_mad_u64_u32 %14, s[24:25], %16, %17, %14\n v_mad_u64_u32 %15, s[26:27], %16, %17, %15\n")
KERNEL(k_mad64_vs_acc, "v_mad_u64_u32 %8, s[20:21], %16, %18, %8\n v_mad_u64_u32 %9, s[22:23], %16, %18, %9\n v_mad_u64_u32 %10, s[24:25], %16, %18, %10\n v_mad_u64_u32 %11, s[26:27], %16, %18, %11\n v_mad_u64_u32 %12, s[20:21], %16, %18, %12\n v_mad_u64_u32 %13, s[22:23], %16, %18, %13\n v_mad_u64_u32 %14, s[24:25], %16, %18, %14\n v_mad_u64_u32 %15, s[26:27], %16, %18, %15\n")
// what the compiler-generated address / glue code adds around the streams
KERNEL(k_lshl_add_u64, "v_lshl_add_u64 %8, %8, 0, %15\n v_lshl_add_u64 %9, %9, 0, %15\n v_lshl_add_u64 %10, %10, 0, %15\n v_lshl_add_u64 %11, %11, 0, %15\n v_lshl_add_u64 %12, %12, 0, %15\n v_lshl_add_u64 %13, %13, 0, %15\n v_lshl_add_u64 %14, %14, 0, %15\n v_lshl_add_u64 %8, %8, 0, %14\n")
KERNEL(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %16\n v_mul_lo_u32 %1, %1, %16\n v_mul_lo_u32 %2, %2, %16\n v_mul_lo_u32 %3, %3, %16\n v_mul_lo_u32 %4, %4, %16\n v_mul_lo_u32 %5, %5, %16\n v_mul_lo_u32 %6, %6, %16\n v_mul_lo_u32 %7, %7, %16\n")
// carry form + the scalar mask op the streams put between two carry steps: does the SALU instruction cost a VALU slot?
KERNEL(k_addc_plus_salu, "v_addc_co_u32 %0, s[20:21], %0, 0, s[20:21]\n s_andn2_b64 s[24:25], s[24:25], s[26:27]\n v_addc_co_u32 %1, s[22:23], %1, 0, s[22:23]\n s_or_b64 s[26:27], s[26:27], s[24:25]\n v_addc_co_u32 %2, s[20:21], %2, 0, s[20:21]\n s_andn2_b64 s[24:25], s[24:25], s[26:27]\n v_addc_co_u32 %3, s[22:23], %3, 0, s[22:23]\n s_or_b64 s[26:27], s[26:27], s[24:25]\n v_addc_co_u32 %4, s[20:21], %4, 0, s[20:21]\n s_andn2_b64 s[24:25], s[24:25], s[26:27]\n v_addc_co_u32 %5, s[22:23], %5, 0, s[22:23]\n s_or_b64 s[26:27], s[26:27], s[24:25]\n v_addc_co_u32 %6, s[20:21], %6, 0, s[20:21]\n s_andn2_b64 s[24:25], s[24:25], s[26:27]\n v_addc_co_u32 %7, s[22:23], %7, 0, s[22:23]\n s_or_b64 s[26:27], s[26:27], s[24:25]\n")

// ---- the generated statements themselves: one radix-8 round (12 butterflies) on 8 register-resident words per iteration
enum { BF_FWD_V = 0, BF_FWD_S = 1, BF_INV_V = 2, BF_INV_S = 3, BF_MUL_V = 4 };

template <int KIND>
__global__ void __launch_bounds__(256) k_round8(uint32_t *out, Stamp *st, int iters, uint32_t seed) {
#if defined(__HIP_DEVICE_COMPILE__)  // gl_asm.h holds device code only; the host pass needs just the symbol
    using namespace ntt;
    const uint64_t P = 0xFFFFFFFF00000001ull;
    uint64_t a[8], t[7];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = ((uint64_t) (threadIdx.x * 2654435761u + seed * (i + 3)) << 21 | (i * 1315423911u)) % P;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const uint64_t v = ((uint64_t) (seed * 40503u + i * 97u) << 29 | (i * 2246822519u + seed)) % P;
        // _s forms: wave-uniform twiddles (SGPRs) as in the column pass; _v forms: per-lane twiddles (VGPRs) as in the first pass
        t[i] = (KIND == BF_FWD_S || KIND == BF_INV_S) ? v : (v + threadIdx.x) % P;
    }
    STAMP_BEGIN
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 3; s++) {
            const int h = 1 << s;  // pairs (j, j + h), twiddle per block as in the network
#pragma unroll
            for (int q = 0; q < 4; q += 2) {
                // two butterflies per statement
                int j0 = ((q >> s) << (s + 1)) | (q & (h - 1));
                int j1 = (((q + 1) >> s) << (s + 1)) | ((q + 1) & (h - 1));
                const uint64_t t0 = t[(4 >> s) - 1 + (j0 >> (s + 1))], t1 = t[(4 >> s) - 1 + (j1 >> (s + 1))];
                if (KIND == BF_FWD_V) gl_fwd2_v(a[j0], a[j0 + h], t0, a[j1], a[j1 + h], t1);
                else if (KIND == BF_FWD_S) gl_fwd2_s(a[j0], a[j0 + h], t0, a[j1], a[j1 + h], t1);
                else if (KIND == BF_INV_V) gl_inv2_v(a[j0], a[j0 + h], t0, a[j1], a[j1 + h], t1);
                else if (KIND == BF_INV_S) gl_inv2_s(a[j0], a[j0 + h], t0, a[j1], a[j1 + h], t1);
                else {
                    gl_mul2_v(a[j0], t0, a[j1], t1);
                    gl_mul2_v(a[j0 + h], t0, a[j1 + h], t1);
                }
            }
        }
    }
    STAMP_END
    uint64_t x = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) x ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t) (x ^ (x >> 32));
#endif
}

// ---- host -----------------------------------------------------------------------------------------------------------
struct Result {
    double cycles_per_instr, ns_per_instr, clock_ghz;
};

template <class K>
Result run(K kern, int waves_per_simd, double instr_per_wave, uint32_t *d_out, Stamp *d_st, std::vector<Stamp> &h_st) {
    const int blocks = CUS * waves_per_simd, threads = 256;
    // exactly `waves_per_simd` workgroups fit a CU: each asks for floor(160 KiB / W) of LDS (64-byte granules kept clear of rounding)
    const size_t lds = (160 * 1024 / waves_per_simd) - 512;
    if (hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds) != hipSuccess) {
        fprintf(stderr, "hipFuncSetAttribute(%zu) failed\n", lds);
        exit(1);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, d_out, d_st, ITERS, 12345u + w);
    hipDeviceSynchronize();
    const int reps = 5;
    std::vector<double> cyc, ghz;
    float ms_total = 0;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, d_out, d_st, ITERS, 777u + r);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        ms_total += ms;
        const size_t nw = (size_t) blocks * threads / 64;
        hipMemcpy(h_st.data(), d_st, nw * sizeof(Stamp), hipMemcpyDeviceToHost);
        std::vector<unsigned long long> c(nw), t(nw);
        for (size_t i = 0; i < nw; i++) c[i] = h_st[i].cycles, t[i] = h_st[i].ticks;
        std::nth_element(c.begin(), c.begin() + nw / 2, c.end());
        std::nth_element(t.begin(), t.begin() + nw / 2, t.end());
        cyc.push_back((double) c[nw / 2]);
        ghz.push_back((double) c[nw / 2] / ((double) t[nw / 2] * 10.0));  // 100 MHz ticks -> ns
    }
    if (hipGetLastError() != hipSuccess) {
        fprintf(stderr, "launch failed\n");
        exit(1);
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(ghz.begin(), ghz.end());
    Result res;
    res.cycles_per_instr = cyc[reps / 2] / (waves_per_simd * instr_per_wave);
    res.clock_ghz = ghz[reps / 2];
    res.ns_per_instr = (ms_total / reps) * 1e6 / (waves_per_simd * instr_per_wave);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return res;
}

int main() {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        fprintf(stderr, "no GPU\n");
        return 1;
    }
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    if (prop.multiProcessorCount != CUS) fprintf(stderr, "warning: %d CUs, the occupancy forcing assumes %d\n", prop.multiProcessorCount, CUS);
    uint32_t *d_out;
    Stamp *d_st;
    const size_t max_threads = (size_t) CUS * 8 * 256;
    hipMalloc(&d_out, max_threads * sizeof(uint32_t));
    hipMalloc(&d_st, max_threads / 64 * sizeof(Stamp));
    std::vector<Stamp> h_st(max_threads / 64);

    printf("{\n \"device\": \"%s\", \"gcn_arch\": \"%s\", \"compute_units\": %d,\n", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    printf(" \"caveat\": \"the cycles figures are the MEDIAN WAVE of a ONE-generation launch divided by an assumed co-residency: a per-wave figure, NOT "
           "the unit's throughput (tools/valu_peak.hip measures that: 4 cycles per VOP3-class form, 2 per plain move / add; this file's own ns column agrees)\",\n");
    printf(" \"method\": \"cycles = median over waves of delta s_memtime around the loop / (waves per SIMD x instructions per wave); "
           "clock_GHz = delta s_memtime / delta s_memrealtime (100 MHz); occupancy forced by LDS size; %d iterations x 64 instructions "
           "(forms) or x 12 butterflies (streams)\",\n", ITERS);
    printf(" \"forms\": {\n");
    struct Form {
        const char *name, *klass;
        void (*k)(uint32_t *, Stamp *, int, uint32_t);
        double instr_per_iter;
    };
    const Form forms[] = {
        {"v_mov_b32", "plain", k_mov_b32, 64},
        {"v_add_u32", "plain", k_add_u32, 64},
        {"v_add_co_u32 (SGPR-pair carry out)", "carry", k_add_co_sgpr, 64},
        {"v_addc_co_u32 (SGPR-pair carry in/out)", "carry", k_addc_co_sgpr, 64},
        {"v_addc_co_u32 x, 0 (SGPR-pair carry)", "carry", k_addc_co_zero, 64},
        {"v_sub_co_u32 (SGPR-pair borrow out)", "carry", k_sub_co_sgpr, 64},
        {"v_subb_co_u32 (SGPR-pair borrow in/out)", "carry", k_subb_co_sgpr, 64},
        {"v_subbrev_co_u32 0, x (SGPR-pair borrow)", "carry", k_subbrev_co_zero, 64},
        {"v_cmp_le_u64 sgpr-pair, s[p], v[pair]", "cmp64", k_cmp_le_u64_sgpr, 64},
        {"v_cndmask_b32 0, 1, sgpr-pair", "cndmask", k_cndmask_01_sgpr, 64},
        {"v_mad_u64_u32 v, v, 0", "mad64", k_mad64_vv_0, 64},
        {"v_mad_u64_u32 v, v, v[pair]", "mad64", k_mad64_vv_acc, 64},
        {"v_mad_u64_u32 v, s, v[pair]", "mad64_s", k_mad64_vs_acc, 64},
        {"v_lshl_add_u64", "other", k_lshl_add_u64, 64},
        {"v_mul_lo_u32", "other", k_mul_lo_u32, 64},
        {"v_addc_co_u32 + s_andn2/s_or_b64 interleaved (per VALU instruction)", "carry+salu", k_addc_plus_salu, 64},
    };
    const int nforms = (int) (sizeof(forms) / sizeof(forms[0]));
    for (int f = 0; f < nforms; f++) {
        printf("  \"%s\": {\"class\": \"%s\", \"cycles\": {", forms[f].name, forms[f].klass);
        std::string ns = "", ghz = "";
        for (int w = 1, first = 1; w <= 8; w += 1, first = 0) {
            Result r = run(forms[f].k, w, forms[f].instr_per_iter * ITERS, d_out, d_st, h_st);
            printf("%s\"%d\": %.3f", first ? "" : ", ", w, r.cycles_per_instr);
            char buf[64];
            snprintf(buf, sizeof(buf), "%s\"%d\": %.3f", first ? "" : ", ", w, r.ns_per_instr);
            ns += buf;
            snprintf(buf, sizeof(buf), "%s\"%d\": %.3f", first ? "" : ", ", w, r.clock_ghz);
            ghz += buf;
            fprintf(stderr, "%-64s W=%d  %.3f cyc  %.3f ns  %.3f GHz\n", forms[f].name, w, r.cycles_per_instr, r.ns_per_instr, r.clock_ghz);
        }
        printf("}, \"ns\": {%s}, \"clock_GHz\": {%s}}%s\n", ns.c_str(), ghz.c_str(), f + 1 < nforms ? "," : "");
    }
    printf(" },\n \"streams\": {\n");
    struct Stream {
        const char *name;
        void (*k)(uint32_t *, Stamp *, int, uint32_t);
        double valu_per_butterfly;
    };
    const Stream streams[] = {
        {"gl_fwd2_v (forward butterfly, per-lane twiddles)", k_round8<BF_FWD_V>, 22},
        {"gl_fwd2_s (forward butterfly, wave-uniform twiddles)", k_round8<BF_FWD_S>, 22},
        {"gl_inv2_v (inverse butterfly, per-lane twiddles)", k_round8<BF_INV_V>, 21},
        {"gl_inv2_s (inverse butterfly, wave-uniform twiddles)", k_round8<BF_INV_S>, 21},
        {"gl_mul2_v (product alone, two per butterfly slot)", k_round8<BF_MUL_V>, 26},
    };
    const int nstreams = (int) (sizeof(streams) / sizeof(streams[0]));
    for (int f = 0; f < nstreams; f++) {
        printf("  \"%s\": {\"valu_per_butterfly\": %.0f, \"cycles_per_butterfly\": {", streams[f].name, streams[f].valu_per_butterfly);
        std::string ghz = "";
        for (int w = 1, first = 1; w <= 4; w += 1, first = 0) {  // 128 VGPRs: at most 4 waves per SIMD, like the kernels (3.7 measured)
            Result r = run(streams[f].k, w, 12.0 * ITERS, d_out, d_st, h_st);
            printf("%s\"%d\": %.2f", first ? "" : ", ", w, r.cycles_per_instr);
            char buf[64];
            snprintf(buf, sizeof(buf), "%s\"%d\": %.3f", first ? "" : ", ", w, r.clock_ghz);
            ghz += buf;
            fprintf(stderr, "%-64s W=%d  %.2f cyc per butterfly  %.3f GHz\n", streams[f].name, w, r.cycles_per_instr, r.clock_ghz);
        }
        printf("}, \"clock_GHz\": {%s}}%s\n", ghz.c_str(), f + 1 < nstreams ? "," : "");
    }
    printf(" }\n}\n");
    hipFree(d_out);
    hipFree(d_st);
    return 0;
}
