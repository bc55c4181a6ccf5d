// valu_peak.hip -- the vector ALU's THROUGHPUT per instruction form, measured the way a roofline needs it: a launch of MANY
// generations of workgroups (a finished wave is replaced at once, as in a pass kernel), cycles per wave-instruction per SIMD =
// launch duration (hipEvents) x shader clock (in-kernel s_memtime / s_memrealtime, median over waves) / wave-instructions per SIMD.
// No assumption about how many waves are resident enters the figure.
//
// Why (round 5): tools/valu_issue_cost.hip divides the MEDIAN WAVE's own cycle count of a ONE-generation launch by (waves per SIMD x
// instructions) -- that presumes all W x 4 waves of a CU run side by side from start to end.  Its own wall-clock column says otherwise
// (r05_valu_issue_cost.json: 1.99 "cycles" but 1.79 ns = 4.3 cycles per VOP3 form at W = 8), and so does round 1's wall-clock table
// (r01_microbench_valu_rates.txt: 1.74 ns).  Two sanity rows pin the method to the chip's data sheet: v_pk_fma_f32 must come out near
// 157.3 TFLOP/s (the published FP32 vector peak), v_fma_f32 shows what a NON-packed wave64 instruction costs.
//
// build: hipcc -O3 --offload-arch=gfx950 tools/valu_peak.hip -o tools/valu_peak        run: tools/valu_peak > profiles/rNN_valu_peak.txt
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <algorithm>
#include <vector>

constexpr int CUS = 256;
constexpr int ITERS = 400;  // x 64 instructions per wave
constexpr int GENS = 12;    // generations of workgroups per launch (8 resident 256-thread workgroups per CU at most)

struct Stamp {
    unsigned long long cycles, ticks;
};

#define REP8(x) x x x x x x x x
#define KERNEL(NAME, BODY)                                                                                             \
    __global__ void __launch_bounds__(256) NAME(uint32_t *out, Stamp *st, int iters, uint32_t seed) {                  \
        uint32_t a = threadIdx.x * 2654435761u + seed, b = threadIdx.x * 40503u + 7 * seed;                            \
        uint64_t r0_ = a, r1_ = b, r2_ = a ^ b, r3_ = a + b, r4_ = 5 + a, r5_ = 6 + b, r6_ = 7 * a, r7_ = 8 * b;       \
        uint32_t w0 = a, w1 = b, w2 = a ^ b, w3 = a + b, w4 = 1 + a, w5 = 2 + b, w6 = 3 * a, w7 = 4 * b;               \
        const uint32_t sg = seed * 77u + 1;                                                                            \
        const unsigned long long c0 = __builtin_amdgcn_s_memtime();                                                    \
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();                                                \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                            \
        for (int i = 0; i < iters; ++i) {                                                                              \
            REP8(asm volatile(BODY                                                                                     \
                              : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7),        \
                                "+v"(r0_), "+v"(r1_), "+v"(r2_), "+v"(r3_), "+v"(r4_), "+v"(r5_), "+v"(r6_), "+v"(r7_) \
                              : "v"(a), "v"(b), "s"(sg)                                                                \
                              : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)                \
        }                                                                                                              \
        const unsigned long long c1 = __builtin_amdgcn_s_memtime();                                                    \
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();                                                \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                            \
        if ((threadIdx.x & 63) == 0) st[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = Stamp{c1 - c0, t1 - t0};      \
        out[blockIdx.x * blockDim.x + threadIdx.x] =                                                                   \
            w0 ^ w1 ^ w2 ^ w3 ^ w4 ^ w5 ^ w6 ^ w7 ^ (uint32_t) (r0_ ^ r1_ ^ r2_ ^ r3_ ^ r4_ ^ r5_ ^ r6_ ^ r7_);         \
    }

// 8 independent register sets, 8 instructions per asm statement, 64 per loop iteration
KERNEL(k_fma_f32, "v_fma_f32 %0, %16, %17, %0\n v_fma_f32 %1, %16, %17, %1\n v_fma_f32 %2, %16, %17, %2\n v_fma_f32 %3, %16, %17, %3\n v_fma_f32 %4, %16, %17, %4\n v_fma_f32 %5, %16, %17, %5\n v_fma_f32 %6, %16, %17, %6\n v_fma_f32 %7, %16, %17, %7\n")
// round 6: is v_fma_f32's 3.28 cycles -- where MI355X_MICROARCH.md's constants table says 2 -- a property of the
// unit or of this probe's operand form (VOP3 with THREE different VGPR sources)?  The VOP2 two-source form v_fmac_f32 (D += S0 * S1),
// a VOP3 v_fma_f32 that reads only TWO different VGPRs, and the VOP2 carry form v_add_co_u32 ..., vcc answer it.
KERNEL(k_fmac_f32, "v_fmac_f32 %0, %16, %17\n v_fmac_f32 %1, %16, %17\n v_fmac_f32 %2, %16, %17\n v_fmac_f32 %3, %16, %17\n v_fmac_f32 %4, %16, %17\n v_fmac_f32 %5, %16, %17\n v_fmac_f32 %6, %16, %17\n v_fmac_f32 %7, %16, %17\n")
KERNEL(k_fma_f32_2src, "v_fma_f32 %0, %16, %16, %0\n v_fma_f32 %1, %16, %16, %1\n v_fma_f32 %2, %16, %16, %2\n v_fma_f32 %3, %16, %16, %3\n v_fma_f32 %4, %16, %16, %4\n v_fma_f32 %5, %16, %16, %5\n v_fma_f32 %6, %16, %16, %6\n v_fma_f32 %7, %16, %16, %7\n")
KERNEL(k_mul_f32, "v_mul_f32 %0, %0, %16\n v_mul_f32 %1, %1, %16\n v_mul_f32 %2, %2, %16\n v_mul_f32 %3, %3, %16\n v_mul_f32 %4, %4, %16\n v_mul_f32 %5, %5, %16\n v_mul_f32 %6, %6, %16\n v_mul_f32 %7, %7, %16\n")
KERNEL(k_add_co_vcc, "v_add_co_u32_e32 %0, vcc, %0, %16\n v_add_co_u32_e32 %1, vcc, %1, %16\n v_add_co_u32_e32 %2, vcc, %2, %16\n v_add_co_u32_e32 %3, vcc, %3, %16\n v_add_co_u32_e32 %4, vcc, %4, %16\n v_add_co_u32_e32 %5, vcc, %5, %16\n v_add_co_u32_e32 %6, vcc, %6, %16\n v_add_co_u32_e32 %7, vcc, %7, %16\n")
KERNEL(k_addc_co_vcc, "v_addc_co_u32_e32 %0, vcc, %0, %16, vcc\n v_addc_co_u32_e32 %1, vcc, %1, %16, vcc\n v_addc_co_u32_e32 %2, vcc, %2, %16, vcc\n v_addc_co_u32_e32 %3, vcc, %3, %16, vcc\n v_addc_co_u32_e32 %4, vcc, %4, %16, vcc\n v_addc_co_u32_e32 %5, vcc, %5, %16, vcc\n v_addc_co_u32_e32 %6, vcc, %6, %16, vcc\n v_addc_co_u32_e32 %7, vcc, %7, %16, vcc\n")
KERNEL(k_pk_fma_f32, "v_pk_fma_f32 %8, %9, %10, %8\n v_pk_fma_f32 %9, %10, %11, %9\n v_pk_fma_f32 %10, %11, %12, %10\n v_pk_fma_f32 %11, %12, %13, %11\n v_pk_fma_f32 %12, %13, %14, %12\n v_pk_fma_f32 %13, %14, %15, %13\n v_pk_fma_f32 %14, %15, %8, %14\n v_pk_fma_f32 %15, %8, %9, %15\n")
KERNEL(k_mov_b32, "v_mov_b32 %0, %16\n v_mov_b32 %1, %17\n v_mov_b32 %2, %16\n v_mov_b32 %3, %17\n v_mov_b32 %4, %16\n v_mov_b32 %5, %17\n v_mov_b32 %6, %16\n v_mov_b32 %7, %17\n")
KERNEL(k_add_u32, "v_add_u32 %0, %0, %16\n v_add_u32 %1, %1, %16\n v_add_u32 %2, %2, %16\n v_add_u32 %3, %3, %16\n v_add_u32 %4, %4, %16\n v_add_u32 %5, %5, %16\n v_add_u32 %6, %6, %16\n v_add_u32 %7, %7, %16\n")
KERNEL(k_add_co_sgpr, "v_add_co_u32 %0, s[20:21], %0, %16\n v_add_co_u32 %1, s[22:23], %1, %16\n v_add_co_u32 %2, s[24:25], %2, %16\n v_add_co_u32 %3, s[26:27], %3, %16\n v_add_co_u32 %4, s[20:21], %4, %16\n v_add_co_u32 %5, s[22:23], %5, %16\n v_add_co_u32 %6, s[24:25], %6, %16\n v_add_co_u32 %7, s[26:27], %7, %16\n")
KERNEL(k_addc_co_sgpr, "v_addc_co_u32 %0, s[20:21], %0, %16, s[20:21]\n v_addc_co_u32 %1, s[22:23], %1, %16, s[22:23]\n v_addc_co_u32 %2, s[24:25], %2, %16, s[24:25]\n v_addc_co_u32 %3, s[26:27], %3, %16, s[26:27]\n v_addc_co_u32 %4, s[20:21], %4, %16, s[20:21]\n v_addc_co_u32 %5, s[22:23], %5, %16, s[22:23]\n v_addc_co_u32 %6, s[24:25], %6, %16, s[24:25]\n v_addc_co_u32 %7, s[26:27], %7, %16, s[26:27]\n")
KERNEL(k_mad64, "v_mad_u64_u32 %8, s[20:21], %16, %17, %8\n v_mad_u64_u32 %9, s[22:23], %16, %17, %9\n v_mad_u64_u32 %10, s[24:25], %16, %17, %10\n v_mad_u64_u32 %11, s[26:27], %16, %17, %11\n v_mad_u64_u32 %12, s[20:21], %16, %17, %12\n v_mad_u64_u32 %13, s[22:23], %16, %17, %13\n v_mad_u64_u32 %14, s[24:25], %16, %17, %14\n v_mad_u64_u32 %15, s[26:27], %16, %17, %15\n")
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %16\n v_mul_lo_u32 %1, %1, %16\n v_mul_lo_u32 %2, %2, %16\n v_mul_lo_u32 %3, %3, %16\n v_mul_lo_u32 %4, %4, %16\n v_mul_lo_u32 %5, %5, %16\n v_mul_lo_u32 %6, %6, %16\n v_mul_lo_u32 %7, %7, %16\n")
KERNEL(k_cndmask_sgpr, "v_cndmask_b32 %0, 0, 1, s[20:21]\n v_cndmask_b32 %1, 0, 1, s[22:23]\n v_cndmask_b32 %2, 0, 1, s[24:25]\n v_cndmask_b32 %3, 0, 1, s[26:27]\n v_cndmask_b32 %4, 0, 1, s[20:21]\n v_cndmask_b32 %5, 0, 1, s[22:23]\n v_cndmask_b32 %6, 0, 1, s[24:25]\n v_cndmask_b32 %7, 0, 1, s[26:27]\n")

int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int blocks = CUS * 8 * GENS;
    uint32_t *d_out;
    Stamp *d_st;
    hipMalloc(&d_out, (size_t) blocks * 256 * sizeof(uint32_t));
    hipMalloc(&d_st, (size_t) blocks * 4 * sizeof(Stamp));
    std::vector<Stamp> h((size_t) blocks * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    struct F {
        const char *name;
        void (*k)(uint32_t *, Stamp *, int, uint32_t);
        double flop_per_lane;  // for the data-sheet rows
    };
    const F forms[] = {{"v_pk_fma_f32 (2 FMA per lane: the data sheet's 157.3 TFLOP/s form)", k_pk_fma_f32, 4},
                       {"v_fma_f32 (one FMA per lane; VOP3, three different VGPR sources)", k_fma_f32, 2},
                       {"v_fma_f32_2src (VOP3, two different VGPR sources: S0 = S1, acc)", k_fma_f32_2src, 2},
                       {"v_fmac_f32 (VOP2: D += S0 * S1, two sources + the accumulator)", k_fmac_f32, 2},
                       {"v_mul_f32 (VOP2, two sources)", k_mul_f32, 1},
                       {"v_add_co_u32_e32 (VOP2, carry out to VCC)", k_add_co_vcc, 0},
                       {"v_addc_co_u32_e32 (VOP2, carry in / out through VCC)", k_addc_co_vcc, 0},
                       {"v_mov_b32", k_mov_b32, 0},
                       {"v_add_u32", k_add_u32, 0},
                       {"v_add_co_u32 (SGPR-pair carry out)", k_add_co_sgpr, 0},
                       {"v_addc_co_u32 (SGPR-pair carry in / out)", k_addc_co_sgpr, 0},
                       {"v_mad_u64_u32 v, v, v[pair]", k_mad64, 0},
                       {"v_mul_lo_u32", k_mul_lo, 0},
                       {"v_cndmask_b32 0, 1, sgpr-pair", k_cndmask_sgpr, 0}};
    printf("# %s, %d CUs; %d workgroups of 256 threads per launch (%d generations of 8 per CU), %d x 64 instructions per wave\n", prop.name,
           prop.multiProcessorCount, blocks, GENS, ITERS);
    printf("# cycles per wave-instruction per SIMD = launch duration x clock / (waves x instructions / 1024 SIMDs); clock = median over waves of\n"
           "# delta s_memtime / delta s_memrealtime (100 MHz)\n");
    printf("%-72s %9s %9s %9s %12s\n", "form", "ns", "clock GHz", "cycles", "TFLOP/s");
    for (const F &f : forms) {
        hipLaunchKernelGGL(f.k, dim3(blocks), dim3(256), 0, 0, d_out, d_st, ITERS, 1u);
        hipDeviceSynchronize();
        std::vector<double> cyc, nss, ghzs;
        for (int r = 0; r < 5; r++) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(f.k, dim3(blocks), dim3(256), 0, 0, d_out, d_st, ITERS, 7u + r);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), d_st, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
            std::vector<double> g(h.size());
            for (size_t i = 0; i < h.size(); i++) g[i] = (double) h[i].cycles / ((double) h[i].ticks * 10.0);
            std::nth_element(g.begin(), g.begin() + g.size() / 2, g.end());
            const double ghz = g[g.size() / 2];
            const double instr_per_simd = (double) blocks * 4 * 64.0 * ITERS / (CUS * 4.0);
            const double ns = ms * 1e6 / instr_per_simd;
            nss.push_back(ns);
            ghzs.push_back(ghz);
            cyc.push_back(ns * ghz);
        }
        if (hipGetLastError() != hipSuccess) return 2;
        std::sort(cyc.begin(), cyc.end());
        std::sort(nss.begin(), nss.end());
        std::sort(ghzs.begin(), ghzs.end());
        char tf[32] = "";
        if (f.flop_per_lane > 0) snprintf(tf, sizeof(tf), "%.1f", f.flop_per_lane * 64.0 / nss[2] * 1e9 * CUS * 4 / 1e12);
        printf("%-72s %9.3f %9.3f %9.2f %12s\n", f.name, nss[2], ghzs[2], cyc[2], tf);
        fflush(stdout);
    }
    return 0;
}
