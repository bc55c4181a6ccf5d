// microbench2.hip -- wall-clock issue cost of candidate VALU instructions on gfx950 at
// 8 waves/SIMD (saturated).  Each kernel runs ITERS x 64 copies of one instruction over 8
// independent register sets.  Output: ns per wave-instruction per SIMD, relative to v_add_u32.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define REP8(x) x x x x x x x x
#define ITERS 2000

#define KERNEL(NAME, BODY)                                                     \
    __global__ void NAME(uint32_t *out, int iters) {                                                      \
        uint32_t a = threadIdx.x * 2654435761u + 1, b = threadIdx.x * 40503u + 7;                         \
        uint64_t r0 = a, r1 = b, r2 = a ^ b, r3 = a + b, r4 = 5, r5 = 6, r6 = 7, r7 = 8;                   \
        uint32_t w0 = a, w1 = b, w2 = a ^ b, w3 = a + b, w4 = 1, w5 = 2, w6 = 3, w7 = 4;                   \
        for (int i = 0; i < iters; ++i) {                                                                 \
            REP8(asm volatile(BODY             \
                              : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6),     \
                                "+v"(w7), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5),     \
                                "+v"(r6), "+v"(r7)                                                        \
                              : "v"(a), "v"(b), "s"(iters)                                                \
                              : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)          \
        }                                                                                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] =                                                      \
            w0 ^ w1 ^ w2 ^ w3 ^ w4 ^ w5 ^ w6 ^ w7 ^ (uint32_t) (r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7);    \
    }
// 32-bit ops on w0..w7 (%0..%7), a=%16 b=%17 sgpr=%18 ; 64-bit regs %8..%15
#define W8(OP, TAIL) OP " %0, %0" TAIL "\n" OP " %1, %1" TAIL "\n" OP " %2, %2" TAIL "\n" OP " %3, %3" TAIL "\n" OP " %4, %4" TAIL "\n" OP " %5, %5" TAIL "\n" OP " %6, %6" TAIL "\n" OP " %7, %7" TAIL "\n"

KERNEL(k_add_u32, W8("v_add_u32", ", %16"))
KERNEL(k_sub_u32, W8("v_sub_u32", ", %16"))
KERNEL(k_and_b32, W8("v_and_b32", ", %16"))
KERNEL(k_xor_b32, W8("v_xor_b32", ", %16"))
KERNEL(k_min_u32, W8("v_min_u32", ", %16"))
KERNEL(k_lshl_b32, W8("v_lshlrev_b32", ", 3"))
KERNEL(k_mov_b32, "v_mov_b32 %0, %16\n" "v_mov_b32 %1, %17\n" "v_mov_b32 %2, %16\n" "v_mov_b32 %3, %17\n" "v_mov_b32 %4, %16\n" "v_mov_b32 %5, %17\n" "v_mov_b32 %6, %16\n" "v_mov_b32 %7, %17\n")
KERNEL(k_add3_u32, W8("v_add3_u32", ", %16, %17"))
KERNEL(k_lshl_add_u32, W8("v_lshl_add_u32", ", 2, %17"))
KERNEL(k_and_or_b32, W8("v_and_or_b32", ", %16, %17"))
KERNEL(k_bfi_b32, W8("v_bfi_b32", ", %16, %17"))
KERNEL(k_alignbit, W8("v_alignbit_b32", ", %16, 7"))
KERNEL(k_perm_b32, W8("v_perm_b32", ", %16, %17"))
KERNEL(k_add_sgpr, W8("v_add_u32", ", %18"))
KERNEL(k_mul_u32_u24, W8("v_mul_u32_u24", ", %16"))
KERNEL(k_mul_hi_u32_u24, W8("v_mul_hi_u32_u24", ", %16"))
KERNEL(k_mad_u32_u24, W8("v_mad_u32_u24", ", %16, %17"))
KERNEL(k_mul_lo_u32, W8("v_mul_lo_u32", ", %16"))
KERNEL(k_mul_hi_u32, W8("v_mul_hi_u32", ", %16"))
KERNEL(k_add_co_vcc, "v_add_co_u32 %0, vcc, %0, %16\n" "v_add_co_u32 %1, vcc, %1, %16\n" "v_add_co_u32 %2, vcc, %2, %16\n" "v_add_co_u32 %3, vcc, %3, %16\n" "v_add_co_u32 %4, vcc, %4, %16\n" "v_add_co_u32 %5, vcc, %5, %16\n" "v_add_co_u32 %6, vcc, %6, %16\n" "v_add_co_u32 %7, vcc, %7, %16\n")
KERNEL(k_add_co_sgpr, "v_add_co_u32 %0, s[20:21], %0, %16\n" "v_add_co_u32 %1, s[22:23], %1, %16\n" "v_add_co_u32 %2, s[24:25], %2, %16\n" "v_add_co_u32 %3, s[26:27], %3, %16\n" "v_add_co_u32 %4, s[20:21], %4, %16\n" "v_add_co_u32 %5, s[22:23], %5, %16\n" "v_add_co_u32 %6, s[24:25], %6, %16\n" "v_add_co_u32 %7, s[26:27], %7, %16\n")
KERNEL(k_addc_vcc, "v_addc_co_u32 %0, vcc, %0, %16, vcc\n" "v_addc_co_u32 %1, vcc, %1, %16, vcc\n" "v_addc_co_u32 %2, vcc, %2, %16, vcc\n" "v_addc_co_u32 %3, vcc, %3, %16, vcc\n" "v_addc_co_u32 %4, vcc, %4, %16, vcc\n" "v_addc_co_u32 %5, vcc, %5, %16, vcc\n" "v_addc_co_u32 %6, vcc, %6, %16, vcc\n" "v_addc_co_u32 %7, vcc, %7, %16, vcc\n")
KERNEL(k_cndmask_vcc, "v_cndmask_b32 %0, %0, %16, vcc\n" "v_cndmask_b32 %1, %1, %16, vcc\n" "v_cndmask_b32 %2, %2, %16, vcc\n" "v_cndmask_b32 %3, %3, %16, vcc\n" "v_cndmask_b32 %4, %4, %16, vcc\n" "v_cndmask_b32 %5, %5, %16, vcc\n" "v_cndmask_b32 %6, %6, %16, vcc\n" "v_cndmask_b32 %7, %7, %16, vcc\n")
KERNEL(k_cmp_lt_u32_vcc, "v_cmp_lt_u32 vcc, %0, %16\n" "v_cmp_lt_u32 vcc, %1, %16\n" "v_cmp_lt_u32 vcc, %2, %16\n" "v_cmp_lt_u32 vcc, %3, %16\n" "v_cmp_lt_u32 vcc, %4, %16\n" "v_cmp_lt_u32 vcc, %5, %16\n" "v_cmp_lt_u32 vcc, %6, %16\n" "v_cmp_lt_u32 vcc, %7, %16\n")
KERNEL(k_mad_u64_u32, "v_mad_u64_u32 %8, vcc, %16, %17, %8\n" "v_mad_u64_u32 %9, vcc, %16, %17, %9\n" "v_mad_u64_u32 %10, vcc, %16, %17, %10\n" "v_mad_u64_u32 %11, vcc, %16, %17, %11\n" "v_mad_u64_u32 %12, vcc, %16, %17, %12\n" "v_mad_u64_u32 %13, vcc, %16, %17, %13\n" "v_mad_u64_u32 %14, vcc, %16, %17, %14\n" "v_mad_u64_u32 %15, vcc, %16, %17, %15\n")
KERNEL(k_mad_u64_u32_c0, "v_mad_u64_u32 %8, vcc, %16, %0, 0\n" "v_mad_u64_u32 %9, vcc, %16, %1, 0\n" "v_mad_u64_u32 %10, vcc, %16, %2, 0\n" "v_mad_u64_u32 %11, vcc, %16, %3, 0\n" "v_mad_u64_u32 %12, vcc, %16, %4, 0\n" "v_mad_u64_u32 %13, vcc, %16, %5, 0\n" "v_mad_u64_u32 %14, vcc, %16, %6, 0\n" "v_mad_u64_u32 %15, vcc, %16, %7, 0\n")
KERNEL(k_lshl_add_u64, "v_lshl_add_u64 %8, %8, 0, %15\n" "v_lshl_add_u64 %9, %9, 0, %15\n" "v_lshl_add_u64 %10, %10, 0, %15\n" "v_lshl_add_u64 %11, %11, 0, %15\n" "v_lshl_add_u64 %12, %12, 0, %15\n" "v_lshl_add_u64 %13, %13, 0, %15\n" "v_lshl_add_u64 %14, %14, 0, %15\n" "v_lshl_add_u64 %8, %8, 0, %14\n")
KERNEL(k_lshrrev_b64, "v_lshrrev_b64 %8, 3, %8\n" "v_lshrrev_b64 %9, 3, %9\n" "v_lshrrev_b64 %10, 3, %10\n" "v_lshrrev_b64 %11, 3, %11\n" "v_lshrrev_b64 %12, 3, %12\n" "v_lshrrev_b64 %13, 3, %13\n" "v_lshrrev_b64 %14, 3, %14\n" "v_lshrrev_b64 %15, 3, %15\n")
KERNEL(k_mov_b64, "v_mov_b64 %8, %9\n" "v_mov_b64 %9, %10\n" "v_mov_b64 %10, %11\n" "v_mov_b64 %11, %12\n" "v_mov_b64 %12, %13\n" "v_mov_b64 %13, %14\n" "v_mov_b64 %14, %15\n" "v_mov_b64 %15, %8\n")
KERNEL(k_fma_f32, W8("v_fma_f32", ", %16, %17"))
KERNEL(k_pk_fma_f32, "v_pk_fma_f32 %8, %8, %9, %10\n" "v_pk_fma_f32 %9, %9, %10, %11\n" "v_pk_fma_f32 %10, %10, %11, %12\n" "v_pk_fma_f32 %11, %11, %12, %13\n" "v_pk_fma_f32 %12, %12, %13, %14\n" "v_pk_fma_f32 %13, %13, %14, %15\n" "v_pk_fma_f32 %14, %14, %15, %8\n" "v_pk_fma_f32 %15, %15, %8, %9\n")
KERNEL(k_fma_f64, "v_fma_f64 %8, %8, %9, %10\n" "v_fma_f64 %9, %9, %10, %11\n" "v_fma_f64 %10, %10, %11, %12\n" "v_fma_f64 %11, %11, %12, %13\n" "v_fma_f64 %12, %12, %13, %14\n" "v_fma_f64 %13, %13, %14, %15\n" "v_fma_f64 %14, %14, %15, %8\n" "v_fma_f64 %15, %15, %8, %9\n")
KERNEL(k_pk_add_u16, W8("v_pk_add_u16", ", %16"))
KERNEL(k_dot4_u32_u8, W8("v_dot4_u32_u8", ", %16, %17"))
KERNEL(k_add_dpp, "v_add_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" "v_add_u32_dpp %1, %2, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" "v_add_u32_dpp %2, %3, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" "v_add_u32_dpp %3, %4, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" "v_add_u32_dpp %4, %5, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" "v_add_u32_dpp %5, %6, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" "v_add_u32_dpp %6, %7, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" "v_add_u32_dpp %7, %0, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")

KERNEL(k_cmp_ge_u64_sgpr, "v_cmp_ge_u64 s[20:21], %8, %9\n" "v_cmp_ge_u64 s[22:23], %9, %10\n" "v_cmp_ge_u64 s[24:25], %10, %11\n" "v_cmp_ge_u64 s[26:27], %11, %12\n" "v_cmp_ge_u64 s[20:21], %12, %13\n" "v_cmp_ge_u64 s[22:23], %13, %14\n" "v_cmp_ge_u64 s[24:25], %14, %15\n" "v_cmp_ge_u64 s[26:27], %15, %8\n")
KERNEL(k_cmp_le_u64_sconst, "v_cmp_le_u64 s[20:21], s[26:27], %8\n" "v_cmp_le_u64 s[22:23], s[26:27], %9\n" "v_cmp_le_u64 s[24:25], s[26:27], %10\n" "v_cmp_le_u64 s[20:21], s[26:27], %11\n" "v_cmp_le_u64 s[22:23], s[26:27], %12\n" "v_cmp_le_u64 s[24:25], s[26:27], %13\n" "v_cmp_le_u64 s[20:21], s[26:27], %14\n" "v_cmp_le_u64 s[22:23], s[26:27], %15\n")
KERNEL(k_cndmask_sgpr, "v_cndmask_b32 %0, %0, %16, s[20:21]\n" "v_cndmask_b32 %1, %1, %16, s[22:23]\n" "v_cndmask_b32 %2, %2, %16, s[24:25]\n" "v_cndmask_b32 %3, %3, %16, s[26:27]\n" "v_cndmask_b32 %4, %4, %16, s[20:21]\n" "v_cndmask_b32 %5, %5, %16, s[22:23]\n" "v_cndmask_b32 %6, %6, %16, s[24:25]\n" "v_cndmask_b32 %7, %7, %16, s[26:27]\n")
KERNEL(k_subbrev_sgpr, "v_subbrev_co_u32 %0, s[20:21], 0, %0, s[20:21]\n" "v_subbrev_co_u32 %1, s[22:23], 0, %1, s[22:23]\n" "v_subbrev_co_u32 %2, s[24:25], 0, %2, s[24:25]\n" "v_subbrev_co_u32 %3, s[26:27], 0, %3, s[26:27]\n" "v_subbrev_co_u32 %4, s[20:21], 0, %4, s[20:21]\n" "v_subbrev_co_u32 %5, s[22:23], 0, %5, s[22:23]\n" "v_subbrev_co_u32 %6, s[24:25], 0, %6, s[24:25]\n" "v_subbrev_co_u32 %7, s[26:27], 0, %7, s[26:27]\n")
KERNEL(k_mad_u64_add1, "v_mad_u64_u32 %8, vcc, %0, 1, %8\n" "v_mad_u64_u32 %9, vcc, %1, 1, %9\n" "v_mad_u64_u32 %10, vcc, %2, 1, %10\n" "v_mad_u64_u32 %11, vcc, %3, 1, %11\n" "v_mad_u64_u32 %12, vcc, %4, 1, %12\n" "v_mad_u64_u32 %13, vcc, %5, 1, %13\n" "v_mad_u64_u32 %14, vcc, %6, 1, %14\n" "v_mad_u64_u32 %15, vcc, %7, 1, %15\n")
KERNEL(k_not_b32, "v_not_b32 %0, %0\n" "v_not_b32 %1, %1\n" "v_not_b32 %2, %2\n" "v_not_b32 %3, %3\n" "v_not_b32 %4, %4\n" "v_not_b32 %5, %5\n" "v_not_b32 %6, %6\n" "v_not_b32 %7, %7\n")
KERNEL(k_sub_sgpr, W8("v_subrev_u32", ", %18"))
KERNEL(k_max_u32, W8("v_max_u32", ", %16"))
KERNEL(k_cmp_ne_u32_sgpr, "v_cmp_ne_u32 s[20:21], 0, %0\n" "v_cmp_ne_u32 s[22:23], 0, %1\n" "v_cmp_ne_u32 s[24:25], 0, %2\n" "v_cmp_ne_u32 s[26:27], 0, %3\n" "v_cmp_ne_u32 s[20:21], 0, %4\n" "v_cmp_ne_u32 s[22:23], 0, %5\n" "v_cmp_ne_u32 s[24:25], 0, %6\n" "v_cmp_ne_u32 s[26:27], 0, %7\n")

template <class K>
double run(K kern, uint32_t *d_out) {
    const int blocks = 512, threads = 1024;  // 8 waves per SIMD
    hipEvent_t e0, e1;
    (void) hipEventCreate(&e0);
    (void) hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_out, ITERS);
    (void) hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_out, ITERS);
    (void) hipEventRecord(e1, 0);
    (void) hipDeviceSynchronize();
    float ms = 0;
    (void) hipEventElapsedTime(&ms, e0, e1);
    return (double) ms * 1e6 / ((double) ITERS * 64 * 8);  // ns per wave-instruction per SIMD
}

// power mode (tools/energy_probe.py samples rocm-smi meanwhile): loop ONE kernel for about `seconds`, print its time per
// wave-instruction per SIMD.   microbench2 power <kernel name> <seconds>
template <class K>
void loop_for(K kern, uint32_t *d_out, double seconds, const char *name) {
    double t = run(kern, d_out);  // ns per wave-instruction per SIMD, also the warm-up
    const double launch_s = t * 1e-9 * ITERS * 64 * 8;
    const int launches = (int) (seconds / launch_s) + 1;
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(kern, dim3(512), dim3(1024), 0, 0, d_out, ITERS);
    (void) hipDeviceSynchronize();
    printf("%s %.4f ns per wave-instruction per SIMD, %d launches\n", name, t, launches);
}

int main(int argc, char **argv) {
    uint32_t *d_out;
    (void) hipMalloc(&d_out, 512 * 1024 * 4);
    if (argc >= 4 && !strcmp(argv[1], "power")) {
        const double sec = atof(argv[3]);
#define P(K) if (!strcmp(argv[2], #K)) { loop_for(K, d_out, sec, #K); return 0; }
        P(k_add_u32) P(k_mov_b32) P(k_and_b32) P(k_min_u32) P(k_mul_lo_u32) P(k_mul_hi_u32) P(k_add_co_sgpr) P(k_addc_vcc)
        P(k_cndmask_sgpr) P(k_subbrev_sgpr) P(k_mad_u64_u32) P(k_mad_u64_u32_c0) P(k_lshl_add_u64) P(k_cmp_le_u64_sconst)
        P(k_cmp_ne_u32_sgpr) P(k_fma_f32) P(k_pk_fma_f32) P(k_fma_f64) P(k_add_sgpr)
        fprintf(stderr, "unknown kernel %s\n", argv[2]);
        return 1;
    }
    double base = run(k_add_u32, d_out);
    base = run(k_add_u32, d_out);
#define R(K) { double t = run(K, d_out); printf("%-20s %.3f ns  x%.2f\n", #K, t, t / base); }
    R(k_add_u32) R(k_sub_u32) R(k_and_b32) R(k_xor_b32) R(k_min_u32) R(k_lshl_b32) R(k_mov_b32) R(k_add3_u32)
    R(k_lshl_add_u32) R(k_and_or_b32) R(k_bfi_b32) R(k_alignbit) R(k_perm_b32) R(k_add_sgpr) R(k_mul_u32_u24)
    R(k_mul_hi_u32_u24) R(k_mad_u32_u24) R(k_mul_lo_u32) R(k_mul_hi_u32) R(k_add_co_vcc) R(k_add_co_sgpr)
    R(k_addc_vcc) R(k_cndmask_vcc) R(k_cmp_lt_u32_vcc) R(k_mad_u64_u32) R(k_mad_u64_u32_c0) R(k_lshl_add_u64)
    R(k_lshrrev_b64) R(k_mov_b64) R(k_fma_f32) R(k_pk_fma_f32) R(k_fma_f64) R(k_pk_add_u16) R(k_dot4_u32_u8) R(k_add_dpp)
    R(k_cmp_ge_u64_sgpr) R(k_cmp_le_u64_sconst) R(k_cndmask_sgpr) R(k_subbrev_sgpr) R(k_mad_u64_add1) R(k_not_b32) R(k_sub_sgpr) R(k_max_u32) R(k_cmp_ne_u32_sgpr)
    return 0;
}
