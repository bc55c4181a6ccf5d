/*
 * ntt_hip.h -- C-ABI of libntt_hip.so, the MI355X (gfx950) NTT engine.
 *
 * This is the drop-in boundary for the hot path of hal-lab-u-tokyo/ntt-aie.
 * Citations are file:line under the reference tree.  The reference launches its
 * device graph as
 *     kernel(bo_instr, n_instr, bo_inA, bo_root, bo_outC); run.wait();
 *                                                   (src/test.cpp:159-160, 181-182)
 * whose runtime sequence is sequence(input, root, output) over three
 * memref<N x i32> (src/aie2.py:320-337).  The same three buffers -- input
 * coefficients, twiddle table "root", output -- are what this library takes;
 * (bo_instr, n_instr) are the AIE instruction stream and have no counterpart.
 *
 * Conventions
 *   - plain C: opaque handle, raw device/host pointers, sizes; no C++/torch types.
 *   - every entry point returns int: 0 = ok, negative = NTT_E_* argument/state
 *     error, positive = hipError_t from the runtime.  Nothing throws or aborts
 *     (reference error contract: ERT state != COMPLETED -> message + return 1,
 *     src/test.cpp:162-166): every entry point is a function-try-block
 *     (csrc/guard.h), a failed host allocation comes back as NTT_E_NOMEM.
 *   - ALIGNMENT: every device DATA pointer handed to a transform (d_in, d_out,
 *     d_a, d_b, d_buf) must be 16-byte aligned -- the kernels move 128-bit
 *     vectors.  hipMalloc / torch allocations are (256 bytes); a row view
 *     &buf[b*N] is whenever N * word_bytes is a multiple of 16, i.e. always
 *     except N = 2 with 4-byte words (and N = 1 rows do not exist: logn >= 1).
 *     A misaligned pointer is refused with NTT_E_ARG before any launch.
 *   - device buffers are caller-owned; polynomials are contiguous [batch][N]
 *     words in the reference's element order (natural order in, src/test.cpp:141).
 *     Words are uint32_t (any odd p < 2^32) or uint64_t (any odd p < 2^64; p = 2^64 - 2^32 + 1 takes a faster path).
 *     Coefficients and twiddles must be canonical residues in [0, p) (the
 *     precondition of vector_modadd / vector_modsub, src/aie_core.cc:41-62).
 *   - launches are asynchronous on the caller's hipStream_t (passed as void*;
 *     NULL = default stream).  A plan is immutable after ntt_plan_set_twiddles
 *     and may be shared by host threads; one plan per device.
 *   - no hidden allocation per call: the plan owns its device twiddle copies (and one counter word for
 *     ntt_count_noncanonical, serialised by a mutex: the only entry point that writes plan state).
 *   - the library reads no environment variable that can change a result (NTT_ROCTX=1 only adds ROCTX ranges);
 *     experiment switches exist only in the separate libntt_hip_exp.so build that tools/ load.
 */
#ifndef NTT_HIP_H
#define NTT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ntt_plan *ntt_plan_t;

/* error codes (negative) */
enum {
    NTT_OK = 0,
    NTT_E_ARG = -1,        /* null pointer / size out of range */
    NTT_E_PRIME = -2,      /* p even, p < 3, or p >= 2^32 for 4-byte words */
    NTT_E_LOGN = -3,       /* logn outside [1, NTT_MAX_LOGN] */
    NTT_E_NOTABLE = -4,    /* transform requested before ntt_plan_set_twiddles */
    NTT_E_NOTINVERTIBLE = -5, /* inverse requested but a twiddle is not a unit mod p (0, or shares a factor with a composite p) */
    NTT_E_LAYOUT = -6,     /* NTT_LAYOUT_AIE_BLOCK16 needs N >= 16 */
    NTT_E_RANGE = -7,      /* a twiddle handed to set_twiddles is >= p */
    NTT_E_NODEVICE = -8,   /* no HIP device / device index out of range */
    NTT_E_NOMEM = -9,      /* host memory: a staging buffer (N words) could not be allocated (std::bad_alloc caught at the boundary) */
    NTT_E_INTERNAL = -10   /* any other C++ exception caught at the boundary (never expected; reported instead of terminating the caller) */
};

#define NTT_MAX_LOGN 28

/* element order of the transform-domain buffer */
enum {
    NTT_LAYOUT_NATURAL = 0,
    /* the reference device's output order: 16 blocks of N/16 words, block
     * ans_order[i] holds natural block i (src/test.cpp:69-71, 212-219; caused by the
     * two tile swaps src/aie2.py:192-209, 257-265) */
    NTT_LAYOUT_AIE_BLOCK16 = 1
};

/* library / build identification: returns 10000*major + 100*minor + patch */
int ntt_version(void);
const char *ntt_error_string(int code);
/* number of visible HIP devices (0 if none); never fails */
int ntt_device_count(void);

/* ---- plan -----------------------------------------------------------------
 * Replaces the compile-time constants the reference bakes into every tile call
 * (logN, p, Barrett w/u: src/aie2.py:14-19, 178-306; src/test.cpp:66, 76-77).
 * word_bytes = 4 -> uint32_t words, odd p < 2^32 (Montgomery arithmetic on the
 * device: results are canonical, hence equal to the reference's Barrett words,
 * src/aie_core.cc:27-39, 64-102); word_bytes = 8 -> any odd p < 2^64 (the reference's `%`-based network takes any modulus,
 * src/test.cpp:48-50): p = 2^64-2^32+1 runs the Goldilocks-specific reduction, every other modulus Montgomery with R = 2^64
 * (about 0.74 x the Goldilocks throughput).  Primality is never checked, as in the reference. */
int ntt_plan_create(ntt_plan_t *out, int logn, uint64_t p, int word_bytes, int device);
int ntt_plan_destroy(ntt_plan_t plan);

/* The "root" buffer (bo_root, src/test.cpp:119-120, 137-143, 150): N words,
 * T[0] unused, T[h+i] is the twiddle of block i at the stage with h blocks
 * (src/test.cpp:45).  Copied to the device (and pre-transformed for the
 * arithmetic the kernels use); the inverse table T^-1 is derived here.
 * host_T is a HOST pointer; synchronous. */
int ntt_plan_set_twiddles(ntt_plan_t plan, const void *host_T);

/* Table rule of the reference host (make_roots + modPow + root[0] = 1,
 * src/test.cpp:15-32, 138): w = g^((p-1)/N) with integer division,
 * T[i] = T[i-1]*w mod p.  Fills host_T (N words of the plan's word size). */
int ntt_make_roots(ntt_plan_t plan, uint64_t g, void *host_T);
/* Tables that make the same network a genuine transform:
 *   kind 1: cyclic  T[h+i] = w^(bitrev(i) * N/(2h)),  w = g^((p-1)/N)
 *   kind 2: negacyclic (Longa-Naehrig) T[k] = psi^-bitrev(k), psi = g^((p-1)/(2N))
 * (kind 0 = ntt_make_roots).  Returns NTT_E_ARG when N does not divide the order. */
int ntt_make_table(ntt_plan_t plan, int kind, uint64_t g, void *host_T);

/* The same tables generated ON THE DEVICE (no host table, no upload): one
 * square-and-multiply per entry from w = g^((p-1)/N) (kind 0/1) or psi^-1 (kind 2), forward and
 * inverse table in table form.  Equivalent to ntt_make_table + ntt_plan_set_twiddles. */
int ntt_plan_generate_twiddles(ntt_plan_t plan, int kind, uint64_t g);
/* Read back the plan's table as plain residues (inverse != 0: the T^-1 table). */
int ntt_plan_get_twiddles(ntt_plan_t plan, int inverse, void *host_T);

/* plan introspection (for harnesses): 0 logn, 1 word_bytes, 2 device,
 * 3 number of HBM passes of one forward transform (default decomposition), 4 has-inverse-table,
 * 32 + i: stages in pass i, 64 + i: first stage of pass i (pass 0 is the contiguous one);
 * 6 number of decomposition alternatives, 7 forced alternative (-1 = chosen by batch),
 * 8 the largest number of passes over all alternatives (capacity for ntt_forward_profile),
 * 256 + 16*a + k for alternative a: k = 0 number of passes, k = 1..7 stages in pass k-1,
 * k = 8..14 first stage of pass k-8, k = 15 the smallest batch this alternative is chosen for;
 * 512 + 16*a + k: the kernel variant of pass k of alternative a (0 = the default kernel of that pass shape; 1 = a single-pass
 * size of 2^10..2^12 words on twice the threads, chosen below the batch that fills the device) */
int64_t ntt_plan_info(ntt_plan_t plan, int what);

/* The stage decomposition into HBM passes is chosen at LAUNCH, by batch size, among alternatives fixed at plan creation
 * from (N, word size, modulus class) -- the role of the reference's slab-size rule, where the per-tile slab follows from
 * N and the number of cores (src/aie2.py:21-28).  The tables are decomposition-agnostic, so alternatives cost no device
 * memory and every alternative computes the same words.
 *   ntt_plan_select: the alternative ntt_forward / ntt_inverse / ntt_polymul_negacyclic run for `batch` (>= 0); a product of
 *   `batch` pairs selects ONCE by `batch` and runs every one of its transforms (also the two-operand launches over 2*batch
 *   rows) with that decomposition.
 *   ntt_plan_set_policy: alternative = -1 (default) chooses by batch; k >= 0 pins alternative k.  Plan configuration, like
 *   ntt_plan_set_twiddles: call it before the plan is shared between host threads. */
int ntt_plan_select(ntt_plan_t plan, size_t batch);
int ntt_plan_set_policy(ntt_plan_t plan, int alternative);

/* A copy of `src` on another device (or the same one): tables are copied device-to-device (hipMemcpyPeer over xGMI, no
 * host round trip, no host table needed -- works after ntt_plan_generate_twiddles too).  This is the multi-device leg of
 * the boundary: the reference broadcasts its one table to every tile below the host (src/aie2.py:96-104, object-fifo
 * broadcast of the root buffer) and scatters / gathers the data per tile (src/aie2.py:83-115); a host shards [B][N]
 * over ntt_device_count() devices by cloning one plan per device and launching each shard on its own stream
 * (tests/cxx/multi_device_host.cpp). */
int ntt_plan_clone(ntt_plan_t src, int device, ntt_plan_t *out);

/* ---- transforms ------------------------------------------------------------
 * Forward = the reference network (src/test.cpp:34-60; tile kernels
 * src/aie_core.cc:161-187, 189-361): stage s = 0..logN-1, stride 2^s,
 *   (x, y) -> (x + y, (x - y) * T[N/2^(s+1) + block])  mod p.
 * d_in natural order, d_out in `out_layout`; d_in == d_out allowed (in place).
 * batch polynomials, contiguous. */
int ntt_forward(ntt_plan_t plan, const void *d_in, void *d_out, size_t batch,
                int out_layout, void *stream);

/* Profiling twin of ntt_forward (the reference brackets one kernel iteration with
 * trace events, src/aie_core.cc:129-131, src/aie2.py:168,316): identical launches
 * with a hipEvent recorded on `stream` around every HBM pass; blocks until done.
 * Writes the number of passes to *n_passes and their durations to ms_per_pass[]
 * (capacity max_passes).  The pass count is that of the alternative chosen for THIS batch
 * (ntt_plan_select), which can exceed ntt_plan_info(plan, 3) (the default decomposition): size the
 * array from ntt_plan_info(plan, 8) = the largest pass count over all alternatives (never above 7);
 * a smaller capacity returns NTT_E_ARG with *n_passes set to the count needed. */
int ntt_forward_profile(ntt_plan_t plan, const void *d_in, void *d_out, size_t batch,
                        int out_layout, void *stream, float *ms_per_pass, int max_passes,
                        int *n_passes);

/* Exact inverse of ntt_forward (no reference counterpart; BASELINE configs 3-4):
 * stages logN-1..0, (u, v) -> (u + v/T, u - v/T), then * N^-1 when scale != 0.
 * d_in is in `in_layout` (what ntt_forward produced), d_out natural order. */
int ntt_inverse(ntt_plan_t plan, const void *d_in, void *d_out, size_t batch,
                int in_layout, int scale, void *stream);

/* d_out[i] = d_a[i] * d_b[i] * scale mod p over batch*N words (scale in [0,p),
 * 1 = plain product).  Any of the pointers may alias. */
int ntt_pointwise_mul(ntt_plan_t plan, const void *d_a, const void *d_b, void *d_out,
                      size_t batch, uint64_t scale, void *stream);

/* Negacyclic product c = a*b mod (x^N + 1, p) with a kind-2 table loaded (no reference counterpart; BASELINE config 4):
 * c = Fwd( InvU(a) . InvU(b) . N^-1 ) with the unscaled inverse network InvU.  Sizes N >= 2^7
 * (Goldilocks) / N >= 2^6 (4-byte words) run the column passes (N >= 2^13 only) of both inverse transforms, then ONE
 * fused middle launch per unit of the first pass (last inverse pass of a and of b, word-by-word product, first forward
 * pass: 3 N words of HBM traffic instead of 7 N; for single-pass sizes that launch is the whole product), then the forward
 * column passes (4-byte words: one launch up to N = 2^13); smaller sizes fold the product into the load of the forward
 * transform's first pass.  d_a and d_b are overwritten (scratch); d_out may alias d_a or d_b.  When d_b directly follows
 * d_a in memory (one [2*batch][N] buffer) both operand transforms run as one launch per pass. */
int ntt_polymul_negacyclic(ntt_plan_t plan, void *d_a, void *d_b, void *d_out,
                           size_t batch, void *stream);

/* Precondition check (blocking, diagnostic): how many of the batch*N words are >= p.  The transforms
 * assume canonical residues, as the reference's vector_modadd / vector_modsub do (src/aie_core.cc:41-62);
 * a non-canonical word gives an unspecified (but memory-safe) result: no kernel reads or writes outside the caller's
 * batch*N words whatever they hold (tests/test_emu_asan.py: the kernels' index rules under AddressSanitizer, exact-size buffers). */
int ntt_count_noncanonical(ntt_plan_t plan, const void *d_buf, size_t batch, uint64_t *host_count);

/* Reference-style partial network (test_stage hook, src/test.cpp:55-58, 67):
 * run stages 0..stage only.  Slow path (one launch per stage), for bring-up. */
int ntt_forward_stages(ntt_plan_t plan, const void *d_in, void *d_out, size_t batch,
                       int stage, void *stream);

#ifdef __cplusplus
}
#endif
#endif
