/*
 * ntt_oracle.h -- CPU restatement of the reference's verification path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the timed CPU baseline.  The product path
 * (ntt_aie_amd + libntt_hip.so) never links, imports or calls it.
 *
 * Parity status: PINNED.  The restatement is checked word-for-word against
 *   (1) the KATs of SURVEY.md section 8(c), produced by the literal reference
 *       lines src/test.cpp:15-60, and
 *   (2) oracle/_ref/libntt_ref.so, the literal lines compiled where they lie
 *       (oracle/build_ref.sh), plus the fixtures under tests/golden/ made
 *       from it (tests/golden/make_golden.py).
 *
 * All citations are file:line under /root/reference.
 */
#ifndef NTT_ORACLE_H
#define NTT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Goldilocks prime 2^64 - 2^32 + 1 (BASELINE.json configs 3-5). */
#define NTT_ORACLE_GOLDILOCKS 0xFFFFFFFF00000001ULL

/* src/test.cpp:15-25 `modPow` (recursive square-and-multiply), widened to
 * 64-bit storage with 128-bit products so it is exact for every p < 2^64. */
uint64_t oracle_modpow(uint64_t x, uint64_t n, uint64_t mod);

/* src/test.cpp:27-32 `make_roots` + the caller's `root[0] = 1` (:138):
 * w = g^((p-1)/n) with INTEGER division (no check that n | p-1),
 * roots[0] = 1, roots[i] = roots[i-1] * w mod p.  n = number of entries. */
void oracle_make_roots_u32(uint32_t n, uint32_t *roots, uint32_t p, uint32_t g);
void oracle_make_roots_u64(uint64_t n, uint64_t *roots, uint64_t p, uint64_t g);

/* src/test.cpp:34-60 `ntt`: in-place, stage s = 0.. with stride t = 2^s,
 * h = n / 2^(s+1) blocks, twiddle roots_rev[h + i] per block i,
 *   a[j]   = (v0 + v1) % p
 *   a[j+t] = ((v0 + p - v1) % p * root) % p
 * stops after stage index == `stage` (:55-58); stage = log2(n)-1 is the full
 * network.  Three `%` per butterfly exactly like the reference. */
void oracle_ntt_u32(uint32_t *a, uint32_t n, const uint32_t *roots_rev,
                    uint32_t p, int stage);
void oracle_ntt_u64(uint64_t *a, uint64_t n, const uint64_t *roots_rev,
                    uint64_t p, int stage);

/* Exact inverse of the network above (no reference counterpart, SURVEY 8a
 * row a-ext): stages in reverse order, each butterfly undone as
 *   w = v * T^-1,  x = (u + w)/2,  y = (u - w)/2   (all mod p),
 * i.e. 2^-1 folded per stage, T^-1 by Fermat.  Returns 0, or -1 if a twiddle
 * that is needed is 0 mod p (not invertible). */
int oracle_intt_u32(uint32_t *a, uint32_t n, const uint32_t *roots_rev,
                    uint32_t p);
int oracle_intt_u64(uint64_t *a, uint64_t n, const uint64_t *roots_rev,
                    uint64_t p);

/* Batched drivers: `batch` polynomials of n words, contiguous [batch][n];
 * nthreads <= 1 runs the scalar loop on the calling thread (the reference is
 * single threaded), otherwise one polynomial per OpenMP task. */
void oracle_ntt_batch_u32(uint32_t *a, uint32_t n, size_t batch,
                          const uint32_t *roots_rev, uint32_t p, int nthreads);
void oracle_ntt_batch_u64(uint64_t *a, uint64_t n, size_t batch,
                          const uint64_t *roots_rev, uint64_t p, int nthreads);
int oracle_intt_batch_u32(uint32_t *a, uint32_t n, size_t batch,
                          const uint32_t *roots_rev, uint32_t p, int nthreads);
int oracle_intt_batch_u64(uint64_t *a, uint64_t n, size_t batch,
                          const uint64_t *roots_rev, uint64_t p, int nthreads);

/* c[i] = a[i] * b[i] * scale mod p (pointwise leg of config 4). */
void oracle_pointwise_u32(uint32_t *c, const uint32_t *a, const uint32_t *b,
                          size_t count, uint32_t p, uint32_t scale);
void oracle_pointwise_u64(uint64_t *c, const uint64_t *a, const uint64_t *b,
                          size_t count, uint64_t p, uint64_t scale);

/* src/test.cpp:69-71, 212-219: device block ans_order[i] holds natural-order
 * block i (16 blocks of n/16 words).  dst and src must not overlap. */
void oracle_block16_u32(uint32_t *dst, const uint32_t *src, uint32_t n);
void oracle_block16_u64(uint64_t *dst, const uint64_t *src, uint64_t n);

/* Tables that turn the same network into a genuine transform:
 *  kind 0: make_roots rule (parity mode, above)
 *  kind 1: cyclic      T[h+i] = w^(bitrev_{log2 h}(i) * n/(2h)), w = g^((p-1)/n)
 *  kind 2: negacyclic  T[h+i] = psi^-(bitrev_{log2 h}(i) * n/(2h)... ) in
 *          Longa-Naehrig order: T[k] = psi^-bitrev_{log2 n}(k), psi = g^((p-1)/(2n))
 * Returns 0, or -1 when n does not divide the needed order. */
int oracle_make_table_u64(int kind, uint64_t n, uint64_t *T, uint64_t p, uint64_t g);

/* Schoolbook negacyclic product c = a*b mod (x^n + 1, p); O(n^2), small n only. */
void oracle_negacyclic_schoolbook_u64(uint64_t *c, const uint64_t *a,
                                      const uint64_t *b, uint64_t n, uint64_t p);

/* FNV-1a 64 over bytes (SURVEY 8c KAT hashes: over the uint32 words). */
uint64_t oracle_fnv1a64(const void *data, size_t nbytes);

#ifdef __cplusplus
}
#endif
#endif
