/*
 * ntt_oracle.c -- CPU restatement of the reference's verification path
 * (/root/reference/src/test.cpp:15-60) in plain C, widened to uint32_t /
 * uint64_t storage with uint64_t / unsigned __int128 products.
 *
 * TEST INFRASTRUCTURE ONLY -- see ntt_oracle.h.  Parity status: PINNED
 * (SURVEY.md 8(c) KATs + oracle/_ref + tests/golden fixtures).
 *
 * The loop nests below follow the reference statement by statement; only the
 * integer types differ.  Where the reference's int32 arithmetic would overflow
 * (p > 46340) the widened types keep the mathematical value, which
 * is what "same network, same table, same input => same words" means outside
 * the literal code's validity window.
 */
#include "ntt_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ---- src/test.cpp:15-25  modPow ------------------------------------------ */
uint64_t oracle_modpow(uint64_t x, uint64_t n, uint64_t mod) {
    /* test.cpp:17-23: n == 0 -> 1; odd -> x * modPow(x*x % mod, n/2) % mod;
     * even -> modPow(x*x % mod, n/2).  Same recursion, exact products. */
    uint64_t ret;
    if (n == 0) {
        ret = 1;
    } else if (n % 2 == 1) {
        uint64_t sq = (uint64_t) (((u128) x * x) % mod);
        ret = (uint64_t) (((u128) x * oracle_modpow(sq, n / 2, mod)) % mod);
    } else {
        uint64_t sq = (uint64_t) (((u128) x * x) % mod);
        ret = oracle_modpow(sq, n / 2, mod);
    }
    return ret;
}

/* ---- src/test.cpp:27-32  make_roots (+ root[0] = 1 from :138) ------------ */
void oracle_make_roots_u32(uint32_t n, uint32_t *roots, uint32_t p, uint32_t g) {
    uint32_t w = (uint32_t) oracle_modpow(g, (p - 1) / n, p); /* test.cpp:28 */
    roots[0] = 1;                                             /* test.cpp:138 */
    for (uint32_t i = 1; i < n; i++) {                        /* test.cpp:29-31 */
        roots[i] = (uint32_t) (((uint64_t) roots[i - 1] * w) % p);
    }
}

void oracle_make_roots_u64(uint64_t n, uint64_t *roots, uint64_t p, uint64_t g) {
    uint64_t w = oracle_modpow(g, (p - 1) / n, p);
    roots[0] = 1;
    for (uint64_t i = 1; i < n; i++) {
        roots[i] = (uint64_t) (((u128) roots[i - 1] * w) % p);
    }
}

/* ---- src/test.cpp:34-60  ntt --------------------------------------------- */
void oracle_ntt_u32(uint32_t *a, uint32_t n, const uint32_t *roots_rev,
                    uint32_t p, int stage) {
    uint32_t t = 1;                               /* test.cpp:36 */
    uint32_t j1, j2, h;
    int idx = 0;                                  /* test.cpp:38 */
    for (uint32_t m = n; m > 1; m >>= 1) {        /* test.cpp:39 */
        j1 = 0;
        h = m / 2;
        for (uint32_t i = 0; i < h; i++) {        /* test.cpp:42 */
            j2 = j1 + t - 1;
            for (uint32_t j = j1; j <= j2; j++) { /* test.cpp:44 */
                uint32_t root = roots_rev[h + i]; /* test.cpp:45 */
                uint64_t v0 = a[j];
                uint64_t v1 = a[j + t];
                a[j] = (uint32_t) ((v0 + v1) % p);                   /* :48 */
                a[j + t] = (uint32_t) ((((v0 + p - v1) % p) * root) % p); /* :49-50 */
            }
            j1 += 2 * t;                          /* test.cpp:52 */
        }
        t <<= 1;                                  /* test.cpp:54 */
        if (idx == stage) {                       /* test.cpp:55-57 */
            return;
        }
        idx += 1;
    }
}

void oracle_ntt_u64(uint64_t *a, uint64_t n, const uint64_t *roots_rev,
                    uint64_t p, int stage) {
    uint64_t t = 1;
    uint64_t j1, j2, h;
    int idx = 0;
    for (uint64_t m = n; m > 1; m >>= 1) {
        j1 = 0;
        h = m / 2;
        for (uint64_t i = 0; i < h; i++) {
            j2 = j1 + t - 1;
            for (uint64_t j = j1; j <= j2; j++) {
                uint64_t root = roots_rev[h + i];
                u128 v0 = a[j];
                u128 v1 = a[j + t];
                a[j] = (uint64_t) ((v0 + v1) % p);
                a[j + t] = (uint64_t) ((((v0 + p - v1) % p) * root) % p);
            }
            j1 += 2 * t;
        }
        t <<= 1;
        if (idx == stage) {
            return;
        }
        idx += 1;
    }
}

/* ---- inverse network (SURVEY 8a a-ext; no reference counterpart) ---------- */
static int ilog2_u64(uint64_t n) {
    int l = 0;
    while ((1ULL << l) < n) l++;
    return l;
}

int oracle_intt_u64(uint64_t *a, uint64_t n, const uint64_t *roots_rev, uint64_t p) {
    int logn = ilog2_u64(n);
    uint64_t inv2 = (p + 1) / 2; /* p odd */
    for (int s = logn - 1; s >= 0; s--) {
        uint64_t t = 1ULL << s;
        uint64_t h = n >> (s + 1);
        for (uint64_t i = 0; i < h; i++) {
            uint64_t root = roots_rev[h + i] % p;
            if (root == 0) return -1;
            uint64_t rinv = oracle_modpow(root, p - 2, p);
            uint64_t j1 = i * 2 * t;
            for (uint64_t j = j1; j < j1 + t; j++) {
                u128 u = a[j];
                u128 w = ((u128) a[j + t] * rinv) % p;
                uint64_t x = (uint64_t) ((u + w) % p);
                uint64_t y = (uint64_t) ((u + p - w) % p);
                a[j] = (uint64_t) (((u128) x * inv2) % p);
                a[j + t] = (uint64_t) (((u128) y * inv2) % p);
            }
        }
    }
    return 0;
}

int oracle_intt_u32(uint32_t *a, uint32_t n, const uint32_t *roots_rev, uint32_t p) {
    int logn = ilog2_u64(n);
    uint64_t inv2 = ((uint64_t) p + 1) / 2;
    for (int s = logn - 1; s >= 0; s--) {
        uint32_t t = 1u << s;
        uint32_t h = n >> (s + 1);
        for (uint32_t i = 0; i < h; i++) {
            uint64_t root = roots_rev[h + i] % p;
            if (root == 0) return -1;
            uint64_t rinv = oracle_modpow(root, (uint64_t) p - 2, p);
            uint32_t j1 = i * 2 * t;
            for (uint32_t j = j1; j < j1 + t; j++) {
                uint64_t u = a[j];
                uint64_t w = ((uint64_t) a[j + t] * rinv) % p;
                uint64_t x = (u + w) % p;
                uint64_t y = (u + p - w) % p;
                a[j] = (uint32_t) ((x * inv2) % p);
                a[j + t] = (uint32_t) ((y * inv2) % p);
            }
        }
    }
    return 0;
}

/* ---- batched drivers ------------------------------------------------------ */
void oracle_ntt_batch_u32(uint32_t *a, uint32_t n, size_t batch,
                          const uint32_t *roots_rev, uint32_t p, int nthreads) {
    int stage = ilog2_u64(n) - 1;
    if (nthreads <= 1) {
        for (size_t b = 0; b < batch; b++) oracle_ntt_u32(a + b * n, n, roots_rev, p, stage);
        return;
    }
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (long long b = 0; b < (long long) batch; b++)
        oracle_ntt_u32(a + (size_t) b * n, n, roots_rev, p, stage);
}

void oracle_ntt_batch_u64(uint64_t *a, uint64_t n, size_t batch,
                          const uint64_t *roots_rev, uint64_t p, int nthreads) {
    int stage = ilog2_u64(n) - 1;
    if (nthreads <= 1) {
        for (size_t b = 0; b < batch; b++) oracle_ntt_u64(a + b * n, n, roots_rev, p, stage);
        return;
    }
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (long long b = 0; b < (long long) batch; b++)
        oracle_ntt_u64(a + (size_t) b * n, n, roots_rev, p, stage);
}

int oracle_intt_batch_u32(uint32_t *a, uint32_t n, size_t batch,
                          const uint32_t *roots_rev, uint32_t p, int nthreads) {
    int rc = 0;
    if (nthreads <= 1) {
        for (size_t b = 0; b < batch; b++) rc |= oracle_intt_u32(a + b * n, n, roots_rev, p);
        return rc;
    }
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) reduction(| : rc)
    for (long long b = 0; b < (long long) batch; b++)
        rc |= oracle_intt_u32(a + (size_t) b * n, n, roots_rev, p);
    return rc;
}

int oracle_intt_batch_u64(uint64_t *a, uint64_t n, size_t batch,
                          const uint64_t *roots_rev, uint64_t p, int nthreads) {
    int rc = 0;
    if (nthreads <= 1) {
        for (size_t b = 0; b < batch; b++) rc |= oracle_intt_u64(a + b * n, n, roots_rev, p);
        return rc;
    }
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) reduction(| : rc)
    for (long long b = 0; b < (long long) batch; b++)
        rc |= oracle_intt_u64(a + (size_t) b * n, n, roots_rev, p);
    return rc;
}

/* ---- pointwise ------------------------------------------------------------ */
void oracle_pointwise_u32(uint32_t *c, const uint32_t *a, const uint32_t *b,
                          size_t count, uint32_t p, uint32_t scale) {
    for (size_t i = 0; i < count; i++) {
        uint64_t ab = ((uint64_t) a[i] * b[i]) % p;
        c[i] = (uint32_t) ((ab * scale) % p);
    }
}

void oracle_pointwise_u64(uint64_t *c, const uint64_t *a, const uint64_t *b,
                          size_t count, uint64_t p, uint64_t scale) {
    for (size_t i = 0; i < count; i++) {
        uint64_t ab = (uint64_t) (((u128) a[i] * b[i]) % p);
        c[i] = (uint64_t) (((u128) ab * scale) % p);
    }
}

/* ---- src/test.cpp:69-71, 212-219  block permutation ----------------------- */
static const int ans_order[16] = {0, 2, 1, 3, 8, 10, 9, 11, 4, 6, 5, 7, 12, 14, 13, 15};

void oracle_block16_u32(uint32_t *dst, const uint32_t *src, uint32_t n) {
    uint32_t block_size = n / 16;              /* test.cpp:213 */
    for (int i = 0; i < 16; i++) {             /* test.cpp:214 */
        uint32_t base_i = (uint32_t) ans_order[i] * block_size;
        for (uint32_t j = 0; j < block_size; j++) dst[base_i + j] = src[i * block_size + j];
    }
}

void oracle_block16_u64(uint64_t *dst, const uint64_t *src, uint64_t n) {
    uint64_t block_size = n / 16;
    for (int i = 0; i < 16; i++) {
        uint64_t base_i = (uint64_t) ans_order[i] * block_size;
        for (uint64_t j = 0; j < block_size; j++) dst[base_i + j] = src[i * block_size + j];
    }
}

/* ---- SURVEY F6 tables ------------------------------------------------------ */
static uint64_t bitrev(uint64_t x, int bits) {
    uint64_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1ULL) << (bits - 1 - i);
    return r;
}

int oracle_make_table_u64(int kind, uint64_t n, uint64_t *T, uint64_t p, uint64_t g) {
    int logn = ilog2_u64(n);
    if (kind == 0) {
        oracle_make_roots_u64(n, T, p, g);
        return 0;
    }
    if (kind == 1) {
        if ((p - 1) % n) return -1;
        uint64_t w = oracle_modpow(g, (p - 1) / n, p);
        T[0] = 1;
        for (uint64_t h = 1; h < n; h <<= 1) {
            int lh = ilog2_u64(h);
            for (uint64_t i = 0; i < h; i++)
                T[h + i] = oracle_modpow(w, bitrev(i, lh) * (n / (2 * h)), p);
        }
        return 0;
    }
    if (kind == 2) {
        if ((p - 1) % (2 * n)) return -1;
        uint64_t psi = oracle_modpow(g, (p - 1) / (2 * n), p);
        uint64_t psi_inv = oracle_modpow(psi, p - 2, p);
        for (uint64_t k = 0; k < n; k++) T[k] = oracle_modpow(psi_inv, bitrev(k, logn), p);
        return 0;
    }
    return -1;
}

void oracle_negacyclic_schoolbook_u64(uint64_t *c, const uint64_t *a,
                                      const uint64_t *b, uint64_t n, uint64_t p) {
    for (uint64_t k = 0; k < n; k++) c[k] = 0;
    for (uint64_t i = 0; i < n; i++) {
        for (uint64_t j = 0; j < n; j++) {
            uint64_t prod = (uint64_t) (((u128) (a[i] % p) * (b[j] % p)) % p);
            uint64_t k = i + j;
            if (k < n) {
                c[k] = (uint64_t) (((u128) c[k] + prod) % p);
            } else {
                k -= n;
                c[k] = (uint64_t) (((u128) c[k] + p - prod) % p);
            }
        }
    }
}

uint64_t oracle_fnv1a64(const void *data, size_t nbytes) {
    const unsigned char *d = (const unsigned char *) data;
    uint64_t hsh = 0xcbf29ce484222325ULL;
    for (size_t i = 0; i < nbytes; i++) {
        hsh ^= d[i];
        hsh *= 0x100000001b3ULL;
    }
    return hsh;
}
