// microbench.hip -- issue-rate of the integer VALU instructions the butterflies are
// made of, on the device at hand (gfx950).  Standalone: hipcc -O3 --offload-arch=gfx950
// tools/microbench.hip -o gpurun_out/microbench && ./microbench
// Prints cycles per wave-instruction per SIMD at 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

enum Kind { MAD64 = 0, MUL_LO, MUL_HI, ADD_CO_PAIR, LSHL_ADD_U64, CMP_LT_U64, CNDMASK, ADD_U32, SUB_CO_CHAIN, KINDS };
static const char *NAMES[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_add_co+v_addc_co (pair)",
                              "v_lshl_add_u64", "v_cmp_lt_u64 (to sgpr)", "v_cndmask_b32 (sgpr mask)",
                              "v_add_u32", "v_mad_u64_u32 dependent chain"};

template <int K>
__global__ void bench(uint64_t *out, unsigned long long *cycles, int iters) {
    uint32_t a = threadIdx.x * 2654435761u + 1, b = threadIdx.x * 40503u + 7;
    uint64_t r0 = a, r1 = b, r2 = a ^ b, r3 = a + b, r4 = 5, r5 = 6, r6 = 7, r7 = 8;
    uint32_t w0 = a, w1 = b, w2 = a ^ b, w3 = a + b, w4 = 1, w5 = 2, w6 = 3, w7 = 4;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (K == MAD64) {
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n"
                              "v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                              "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n"
                              "v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                              : "v"(a), "v"(b) : "vcc");)
        } else if constexpr (K == SUB_CO_CHAIN) {
            REP64(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n" : "+v"(r0) : "v"(a), "v"(b) : "vcc");)
        } else if constexpr (K == MUL_LO) {
            REP8(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n"
                              "v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n"
                              "v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                              : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7)
                              : "v"(a));)
        } else if constexpr (K == MUL_HI) {
            REP8(asm volatile("v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n"
                              "v_mul_hi_u32 %3, %3, %8\n v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n"
                              "v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8\n"
                              : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7)
                              : "v"(a));)
        } else if constexpr (K == ADD_CO_PAIR) {
            REP8(asm volatile("v_add_co_u32 %0, vcc, %0, %8\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n"
                              "v_add_co_u32 %2, vcc, %2, %8\n v_addc_co_u32 %3, vcc, %3, %8, vcc\n"
                              "v_add_co_u32 %4, vcc, %4, %8\n v_addc_co_u32 %5, vcc, %5, %8, vcc\n"
                              "v_add_co_u32 %6, vcc, %6, %8\n v_addc_co_u32 %7, vcc, %7, %8, vcc\n"
                              : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7)
                              : "v"(a) : "vcc");)
        } else if constexpr (K == LSHL_ADD_U64) {
            REP8(asm volatile("v_lshl_add_u64 %0, %0, 0, %8\n v_lshl_add_u64 %1, %1, 0, %8\n"
                              "v_lshl_add_u64 %2, %2, 0, %8\n v_lshl_add_u64 %3, %3, 0, %8\n"
                              "v_lshl_add_u64 %4, %4, 0, %8\n v_lshl_add_u64 %5, %5, 0, %8\n"
                              "v_lshl_add_u64 %6, %6, 0, %8\n v_lshl_add_u64 %7, %7, 0, %8\n"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                              : "v"(r7 | 1));)
        } else if constexpr (K == CMP_LT_U64) {
            REP8(asm volatile("v_cmp_lt_u64 s[20:21], %0, %1\n v_cmp_lt_u64 s[22:23], %1, %2\n"
                              "v_cmp_lt_u64 s[24:25], %2, %3\n v_cmp_lt_u64 s[26:27], %3, %4\n"
                              "v_cmp_lt_u64 s[20:21], %4, %5\n v_cmp_lt_u64 s[22:23], %5, %6\n"
                              "v_cmp_lt_u64 s[24:25], %6, %7\n v_cmp_lt_u64 s[26:27], %7, %0\n"
                              : : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7)
                              : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
        } else if constexpr (K == CNDMASK) {
            REP8(asm volatile("v_cndmask_b32 %0, %0, %8, s[20:21]\n v_cndmask_b32 %1, %1, %8, s[20:21]\n"
                              "v_cndmask_b32 %2, %2, %8, s[20:21]\n v_cndmask_b32 %3, %3, %8, s[20:21]\n"
                              "v_cndmask_b32 %4, %4, %8, s[20:21]\n v_cndmask_b32 %5, %5, %8, s[20:21]\n"
                              "v_cndmask_b32 %6, %6, %8, s[20:21]\n v_cndmask_b32 %7, %7, %8, s[20:21]\n"
                              : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7)
                              : "v"(a) : "s20", "s21");)
        } else if constexpr (K == ADD_U32) {
            REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n"
                              "v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n"
                              "v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                              : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7)
                              : "v"(a));)
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ w0 ^ w1 ^ w2 ^ w3 ^ w4 ^ w5 ^ w6 ^ w7;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int K>
void run(uint64_t *d_out, unsigned long long *d_cyc) {
    const int iters = 2000, insts = 64;
    for (int waves_per_simd : {1, 2, 4, 8}) {
        // 1024-thread blocks hold 4 waves per SIMD; 8 per SIMD = two such blocks per CU
        const int threads = waves_per_simd >= 4 ? 1024 : 64 * 4 * waves_per_simd;
        const int blocks = 256 * (waves_per_simd >= 4 ? waves_per_simd / 4 : 1);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(bench<K>, dim3(blocks), dim3(threads), 0, 0, d_out, d_cyc, iters);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(bench<K>, dim3(blocks), dim3(threads), 0, 0, d_out, d_cyc, iters);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[512];
        hipMemcpy(h, d_cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
        double avg = 0;
        for (int i = 0; i < blocks; i++) avg += (double) h[i];
        avg /= blocks;
        // one wave issued iters*insts instructions; waves_per_simd waves share a SIMD
        double cyc_per_inst_per_simd = avg / ((double) iters * insts * waves_per_simd);
        // wall-clock view: every SIMD of the chip retired iters*insts*waves_per_simd wave-instructions
        double ns_per_inst_per_simd = (double) ms * 1e6 / ((double) iters * insts * waves_per_simd);
        printf("%-34s waves/SIMD=%d  wave-time=%9.0f ticks  -> %.2f ticks (%.3f ns wall) per wave-instruction per SIMD"
               "  [kernel %.1f us => %.0f MHz tick]\n", NAMES[K], waves_per_simd, avg, cyc_per_inst_per_simd,
               ns_per_inst_per_simd, ms * 1e3, avg / (ms * 1e3));
    }
}

int main() {
    uint64_t *d_out;
    unsigned long long *d_cyc;
    hipMalloc(&d_out, 512 * 1024 * 8);
    hipMalloc(&d_cyc, 512 * 8);
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    printf("device %s, %d CUs, clock %d kHz (s_memtime ticks at the shader clock)\n", pr.gcnArchName,
           pr.multiProcessorCount, pr.clockRate);
    run<MAD64>(d_out, d_cyc);
    run<SUB_CO_CHAIN>(d_out, d_cyc);
    run<MUL_LO>(d_out, d_cyc);
    run<MUL_HI>(d_out, d_cyc);
    run<ADD_CO_PAIR>(d_out, d_cyc);
    run<LSHL_ADD_U64>(d_out, d_cyc);
    run<CMP_LT_U64>(d_out, d_cyc);
    run<CNDMASK>(d_out, d_cyc);
    run<ADD_U32>(d_out, d_cyc);
    return 0;
}
