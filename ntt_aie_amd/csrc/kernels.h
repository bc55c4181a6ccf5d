// kernels.h -- type-erased launch interface between the C-ABI (ntt_api.hip) and
// the per-field / per-direction kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pass.h"

namespace ntt {

struct ErasedArgs {
    const void *in;
    void *out;
    const void *tw;
    const void *tw_sc;     // scaled inverse, Goldilocks CONTIG pass: stage-0 twiddles * N^-1 (PassArgs::tw_sc); null = phase_scale
    uint32_t p, pinv, r2;  // FieldM32 parameters (ignored by FieldGL)
    uint64_t p64, pinv64, r2_64;  // FieldM64 parameters (general odd 64-bit modulus)
    int n, s0;
    uint32_t batch;
    int layout;
    int do_scale;
    uint64_t scale;  // table form
    uint32_t target_wgs;
    int dbg;  // timing experiments (NTT_DEBUG_FLAGS), 0 in production
    const void *tw2;     // product_mid launch: the FORWARD table (tw is the inverse one there)
    const void *in2;     // forward CONTIG pass: second operand of a fused pointwise product (or null)
    uint64_t pw_scale;   // scale * R^2 (see PassArgs::pw_scale)
    const void *skip_if;  // experiment build only: device word, non-zero = the launch is a no-op (fallback behind the fused kernel)
    int variant;          // PassDesc::variant (plan.h): 0 = the default kernel of this (contig, log_m); 1 = single-pass CONTIG unit of 10..12
                          // stages as radix-8 rounds in 512 threads (twice the waves per unit: small batches, one generation of workgroups)
#if defined(NTT_PHASE_STAMPS)
    void *stamps;            // diagnostic build: PassArgs::stamps / stamp_records (ntt_stamps_set)
    uint32_t stamp_records;
#endif
};

// Each returns hipSuccess / a hipError_t; hipErrorInvalidValue for an
// unsupported (contig, log_m) combination.
hipError_t launch_gl_fwd(bool contig, int log_m, const ErasedArgs &a, hipStream_t s);
hipError_t launch_gl_inv(bool contig, int log_m, const ErasedArgs &a, hipStream_t s);
hipError_t launch_m32_fwd(bool contig, int log_m, const ErasedArgs &a, hipStream_t s);
hipError_t launch_m32_inv(bool contig, int log_m, const ErasedArgs &a, hipStream_t s);
hipError_t launch_m64_fwd(bool contig, int log_m, const ErasedArgs &a, hipStream_t s);  // any odd p < 2^64 (FieldM64)
hipError_t launch_m64_inv(bool contig, int log_m, const ErasedArgs &a, hipStream_t s);

// Fused middle of a negacyclic product (pass.h: run_product_pass): per 2^log_m-word unit, inverse CONTIG pass of a.in
// and of a.in2, word-by-word product * pw_scale, forward CONTIG pass -> a.out.  tw = inverse table, tw2 = forward table.
// hipErrorInvalidValue when this (word size, log_m) has no fused kernel (callers then run the separate passes).
hipError_t launch_gl_product_mid(int log_m, const ErasedArgs &a, hipStream_t s);
bool have_gl_product_mid(int log_m);   // Goldilocks: unit sizes 2^7 .. 2^12
// one grid covers the batch (blockIdx.y range) -- computed by the launcher's own geometry call
bool gl_product_mid_fits(int log_m, int n, uint32_t batch, uint32_t target_wgs);
bool m32_product_mid_fits(int log_m, int n, uint32_t batch, uint32_t target_wgs);
hipError_t launch_m64_product_mid(int log_m, const ErasedArgs &a, hipStream_t s);
bool have_m64_product_mid(int log_m);  // general 64-bit modulus: unit sizes 2^7 .. 2^12, as Goldilocks
bool m64_product_mid_fits(int log_m, int n, uint32_t batch, uint32_t target_wgs);
hipError_t launch_m32_product_mid(int log_m, const ErasedArgs &a, hipStream_t s);
bool have_m32_product_mid(int log_m);  // 4-byte words: unit sizes 2^6 .. 2^13

#if defined(NTT_EXPERIMENT)
// Tools-side experiment, NOT part of libntt_hip.so (tools/fused_gl16.hip, libntt_hip_exp.so only):
// N = 2^16 Goldilocks forward as one persistent XCD-local launch: memset + fused +
// check; must be followed by the ordinary passes with skip_if = fused_gl16_ok_word(ctl).
size_t fused_gl16_ctl_bytes(size_t max_batch);
const void *fused_gl16_ok_word(const void *ctl);
hipError_t launch_fused_gl16(const void *in, void *out, const void *tw, size_t batch, void *ctl_mem, hipStream_t s,
                             int dbg = 0);
#endif

// elementwise c = a*b*scale (scale in plain form; scale == 1 skips the second product)
hipError_t launch_pointwise_gl(const void *a, const void *b, void *c, size_t count, uint64_t scale,
                               hipStream_t s);
hipError_t launch_pointwise_m32(const void *a, const void *b, void *c, size_t count, uint32_t p,
                                uint32_t pinv, uint32_t r2, uint32_t scale, hipStream_t s);
hipError_t launch_pointwise_m64(const void *a, const void *b, void *c, size_t count, uint64_t p,
                                uint64_t pinv, uint64_t r2, uint64_t scale, hipStream_t s);

// device-side table generation (no host upload): T[i] = base^e_kind(i), table form
hipError_t launch_gen_table_gl(void *T, int logn, int kind, uint64_t base_m, uint64_t one_m, hipStream_t s);
hipError_t launch_gen_table_m32(void *T, int logn, int kind, uint32_t base_m, uint32_t one_m, uint32_t p,
                                uint32_t pinv, uint32_t r2, hipStream_t s);
hipError_t launch_gen_table_m64(void *T, int logn, int kind, uint64_t base_m, uint64_t one_m, uint64_t p,
                                uint64_t pinv, uint64_t r2, hipStream_t s);

// out[i] = T[i] * c (table form both): the N/2 scaled stage-0 twiddles of the Goldilocks inverse transform
hipError_t launch_scale_table_gl(const void *T, void *out, size_t count, uint64_t c_m, hipStream_t s);
hipError_t launch_scale_table_m64(const void *T, void *out, size_t count, uint64_t c_m, uint64_t p, uint64_t pinv, uint64_t r2,
                                  hipStream_t s);

// number of words >= p in a buffer (precondition check); d_out = one zeroed 64-bit device word
hipError_t launch_count_noncanonical(const void *a, size_t count, int word_bytes, uint64_t p, void *d_out, hipStream_t s);

// one stage of the network, one thread per butterfly (bring-up path, test_stage hook)
hipError_t launch_stage_gl(void *data, const void *tw, int n, int stage, size_t batch, hipStream_t s);
hipError_t launch_stage_m32(void *data, const void *tw, int n, int stage, size_t batch, uint32_t p,
                            uint32_t pinv, uint32_t r2, hipStream_t s);
hipError_t launch_stage_m64(void *data, const void *tw, int n, int stage, size_t batch, uint64_t p,
                            uint64_t pinv, uint64_t r2, hipStream_t s);

}  // namespace ntt
