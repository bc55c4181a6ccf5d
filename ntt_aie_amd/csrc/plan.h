// plan.h -- host-only planning and table preparation shared by the C-ABI
// (ntt_api.hip) and the host index model (tests/emu).  No HIP types here.
#pragma once
#include <stdint.h>

#include <vector>

namespace ntt {
namespace host {

using u128 = unsigned __int128;

constexpr uint64_t GOLDILOCKS = 0xFFFFFFFF00000001ULL;
constexpr int MAX_CONTIG_LOG_M = 12;  // 4096 words per workgroup tile
constexpr int MIN_COL_LOG_M = 4;
constexpr int MAX_COL_LOG_M = 8;
constexpr int MAX_COL_LOG_M_WIDE = 9;  // 512 rows x one 128-byte segment (one more thread bit): used for N = 2^22 = 13 + 9

struct PassDesc {
    bool contig;
    int s0;
    int log_m;
    int variant = 0;  // which kernel of this (contig, log_m) runs: 0 = default; 1 = wide radix-8 (single-pass CONTIG of 10..12 stages, 512 threads x 8 words)
};

inline uint64_t mulmod(uint64_t a, uint64_t b, uint64_t p) { return (uint64_t) (((u128) a * b) % p); }

inline uint64_t powmod(uint64_t x, uint64_t e, uint64_t p) {
    uint64_t r = 1 % p;
    x %= p;
    while (e) {
        if (e & 1) r = mulmod(r, x, p);
        x = mulmod(x, x, p);
        e >>= 1;
    }
    return r;
}

// a^-1 mod p by the extended Euclidean algorithm; 0 when gcd(a, p) != 1.  (Not Fermat: the reference
// never checks that its modulus is prime, and the 4-byte-word engine accepts any odd p.)
inline uint64_t invmod(uint64_t a, uint64_t p) {
    typedef __int128 i128;
    i128 r0 = (i128) p, r1 = (i128) (a % p), t0 = 0, t1 = 1;
    while (r1 != 0) {
        const i128 q = r0 / r1;
        const i128 r2 = r0 - q * r1, t2 = t0 - q * t1;
        r0 = r1; r1 = r2; t0 = t1; t1 = t2;
    }
    if (r0 != 1) return 0;
    if (t0 < 0) t0 += (i128) p;
    return (uint64_t) t0;
}

inline uint64_t bitrev(uint64_t x, int bits) {
    uint64_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1ULL) << (bits - 1 - i);
    return r;
}

// Split logn stages into HBM passes: one contiguous pass of <= 12 stages (13 for 4-byte words; tile =
// 4096 / 8192 words in LDS) followed by column passes of 4..8 stages (<= 256 rows x one 128-byte segment).
// Always the fewest passes; for the two-pass sizes the split is the measured optimum
// (tools/split_sweep.py, profiles/r01_g_split_sweep.jsonl): a pass costs about max(VALU, memory) plus a
// quarter of the smaller one; a light column pass (6-7 stages) streams at the device's copy rate, and an
// 8-stage first pass is the sweet spot of both CONTIG kernels.
// 4-byte words also have a 13-stage contiguous pass (8192 words = 32 KiB per tile, 512 threads x 16 words, rounds of
// 4 + 4 + 4 + 1 stages; the 8-byte twin would hold one workgroup per CU): N = 2^13 in ONE pass (2.08-2.63 ms per 4 GiB instead of
// 3.19-3.24 in two; a single launch at batch 1: 6.5 instead of 8.1 us) and N = 2^21 in two (13 + 8: 3.80 instead of 4.84 ms);
// 13 + 7 and 13 + 6 lose to 12 + 8 and 12 + 7 (same-process, tools/ab_latency.py with NTT_PLAN_SPLIT).
constexpr int MAX_CONTIG_LOG_M_W4 = 13;

inline std::vector<PassDesc> plan_passes(int n, int word_bytes = 8) {
    std::vector<PassDesc> v;
    if (n <= (word_bytes == 4 ? MAX_CONTIG_LOG_M_W4 : MAX_CONTIG_LOG_M)) {
        v.push_back({true, 0, n});
        return v;
    }
    // ... and both word sizes reach N = 2^21 in two passes, 13 + 8 (8-byte words: 4.40 instead of 4.80 ms per 4 GiB in three; their
    // 13-stage pass re-reads its twiddles from L2 every round (128 VGPRs, two 512-thread workgroups per CU), which at N = 2^13 alone
    // is worth -11 % at saturating batches but +22 % for a single polynomial, so 8-byte N = 2^13 stays 7 + 6)
    // ... and N = 2^22 = 13 + 9: the 9-stage column pass (512 rows per tile, 512 / 1024 threads) saves the third HBM trip
    // (round 3; 8 + 7 + 7 before)
    if (n == MAX_CONTIG_LOG_M_W4 + MAX_COL_LOG_M || n == MAX_CONTIG_LOG_M_W4 + MAX_COL_LOG_M_WIDE) {
        v.push_back({true, 0, MAX_CONTIG_LOG_M_W4});
        v.push_back({false, MAX_CONTIG_LOG_M_W4, n - MAX_CONTIG_LOG_M_W4});
        return v;
    }
    if (n <= MAX_CONTIG_LOG_M + MAX_COL_LOG_M) {
        //                              n = 13  14  15  16  17  18  19  20
        // re-swept on the final kernels of round 2 (tools/split_sweep.py over every split, then same-process confirmation of the
        // candidates: profiles/r02_split_sweep_final.txt): 8-byte 2^15 = 7 + 8 (-1.5 .. -2.2 % against 8 + 7); 4-byte 2^15 = 9 + 6
        // (-0.6 % for a 32-bit prime, -4.7 % for a lazy one), 2^16 = 10 + 6 (+1.5 % / -7.2 %), 2^19 = 11 + 8 (-4.3 % / -1.3 .. -2.3 %)
        static const int first_w8[8] = {7, 8, 7, 8, 9, 10, 11, 12};  // N = 2^16 per 4 GiB (batch 8192): 8 + 8 3.311 ms, 9 + 7 3.339, 10 + 6 3.492 (r02_split_sweep_final.txt)
        static const int first_w4[8] = {8, 8, 9, 10, 10, 10, 11, 12};
        const int first = (word_bytes == 8 ? first_w8 : first_w4)[n - MAX_CONTIG_LOG_M - 1];
        v.push_back({true, 0, first});
        v.push_back({false, first, n - first});
        return v;
    }
    const int extra = (n - MAX_CONTIG_LOG_M + MAX_COL_LOG_M - 1) / MAX_COL_LOG_M;
    const int P = 1 + extra;
    int first = (n + P - 1) / P;
    if (first < n - MAX_COL_LOG_M * extra) first = n - MAX_COL_LOG_M * extra;
    if (first > MAX_CONTIG_LOG_M) first = MAX_CONTIG_LOG_M;
    v.push_back({true, 0, first});
    int rest = n - first, s0 = first;
    for (int i = 0; i < extra; i++) {
        int m = (rest + (extra - i) - 1) / (extra - i);
        v.push_back({false, s0, m});
        s0 += m;
        rest -= m;
    }
    return v;
}

// ---- plan alternatives: the decomposition is chosen at LAUNCH, by batch size, among candidates fixed at plan creation by
// (N, word size, modulus class).  The twiddle tables are direction- and split-agnostic (a pass addresses T by stage and
// block), so an alternative costs no device memory.  What the reference does with its one knob: the slab size follows
// from N and the core count (src/aie2.py:21-28).
//
// Rule: alternatives are ordered by ascending `min_batch`; the launcher takes the LAST one whose min_batch <= batch
// (alternative 0 has min_batch 0).  Thresholds are measured crossovers (profiles/r03_plan_alternatives.txt):
//   * a long single pass (13 / 14 stages in 512 / 1024-thread workgroups, two / one per CU) beats two short passes once the
//     batch fills the device, and loses for a handful of polynomials (round 2: 8-byte N = 2^13 in one pass -11 % at
//     saturating batches, +22 % for one polynomial; 4-byte N = 2^14 with a lazy prime -21 % / +20 %).
struct PlanAlt {
    std::vector<PassDesc> passes;
    uint64_t min_batch;
};

// modulus classes of the 4-byte-word kernels (pass_kernel.inc picks the instruction stream the same way)
inline bool m32_lazy_modulus(uint64_t p) { return p < 0x40000000ull; }

// measured crossovers (same process, tools/ab_latency.py; round 3's driver script is in the history -> profiles/r03_plan_alternatives.txt):
//   8-byte N = 2^13, 7 + 6 -> 13:  batch 1 +23 %, 32 +15 %, 64 +7 %, 128 -7.5 %, 256 -13 %, 2048 -17 %, 65536 -10 % (inverse -6 .. -22 %)
//   4-byte lazy N = 2^14, 8 + 6 -> 14:  batch 1 +18 %, 32 +8 %, 64 0 %, 128 -15 %, 256 -22 %, 1024..4096 -1 .. -3 %, 65536 -20 % (inverse -1 .. -22 %)
#ifndef NTT_ALT_MIN_BATCH_GL13
#define NTT_ALT_MIN_BATCH_GL13 128    // 8-byte N = 2^13: one 13-stage pass from this batch on (512 threads, two workgroups per CU)
#endif
#ifndef NTT_ALT_MIN_BATCH_M32_14
#define NTT_ALT_MIN_BATCH_M32_14 128  // 4-byte lazy primes, N = 2^14: one 14-stage pass (one 1024-thread workgroup per CU)
#endif

// measured crossover of the wide variant (same process, interleaved, outputs compared: profiles/r04_ab_m32_wide.txt; wide against
// radix-16, forward / inverse):
//   p >= 2^31 (carry-select butterflies), N = 2^12: batch 1 -11 / -11 %, 64 -18 / -13 %, 256 -18 / -13 %, 1024 -6 / 0 %, 4096 +3 / +12 %,
//     65536 +3 / +5 %; N = 2^11: 1 .. 256 -5 .. -6 / -12 .. -18 %, 1024 -1 / -9 %, 4096 +4 / +4 %; N = 2^10: 1 .. 1024 -4 .. -6 / -8 .. -12 %
//   p < 2^31 (v_min corrections) and p < 2^30 (lazy): -1 .. -6 % up to batch 256, even or +3 % at 1024, +3 .. +19 % from 4096 on
#ifndef NTT_ALT_MIN_BATCH_M32_WIDE
#define NTT_ALT_MIN_BATCH_M32_WIDE 2048      // 4-byte N = 2^10 .. 2^12, p >= 2^31: the default radix-16 kernel from this batch on
#endif
//   8-byte words (Goldilocks; profiles/r04_ab_gl_wide.txt), N = 2^12: batch 1 -17 / -19 %, 64 -15 / -16 %, 256 -14 / -15 %, 1024 -3 / -1 %,
//     4096 +5 / +6 %, 16384 +6 / +11 %; N = 2^11: 1 .. 256 -12 .. -17 %, 1024 -7 / -7 %, 4096 +4 / -8 %; N = 2^10: 1 .. 1024 -8 .. -16 %, 4096 +4 / -10 %
#ifndef NTT_ALT_MIN_BATCH_W8_WIDE
#define NTT_ALT_MIN_BATCH_W8_WIDE 2048       // 8-byte N = 2^10 .. 2^12: the default radix-16 kernel from this batch on
#endif
#ifndef NTT_ALT_MIN_BATCH_M32_WIDE_LIGHT
#define NTT_ALT_MIN_BATCH_M32_WIDE_LIGHT 512  // ... p < 2^31 (9 / 11-instruction butterflies: the wide variant's extra exchange weighs more)
#endif

inline std::vector<PlanAlt> plan_alternatives(int n, int word_bytes, uint64_t p) {
    std::vector<PlanAlt> alts;
    std::vector<PassDesc> def = plan_passes(n, word_bytes);
    // (4-byte N = 2^16 by modulus class, re-measured in round 3: 10 + 6 against 8 + 8 is -7.1 % for a lazy prime, -1.7 % for a
    // 31-bit one and +0.2 % -- noise -- for a 32-bit one; at batch 1 all within 1 %: one split, plan_passes()'s, for all classes)
    alts.push_back({def, 0});
    if (n == 22 && word_bytes == 4 && !m32_lazy_modulus(p)) {
        // N = 2^22 = 13 + 9 (two HBM trips, the 512-row column tile) against round 2's 8 + 7 + 7, same process, bursts of 20 launches
        // (profiles/r03_n22_classes.txt): Goldilocks -1 .. -5 % at every batch, both directions; lazy 4-byte primes -3 .. -17 % forward,
        // +4 .. +6 % inverse at batches 8 .. 32 and even from 64 on; the heavier 4-byte streams (p >= 2^30) gain only for one or two
        // polynomials (-7 .. -14 %) and lose +5 .. +16 % from batch 8 on (their 13-stage pass is VALU-bound): three light passes there
        alts.push_back({{{true, 0, 8}, {false, 8, 7}, {false, 15, 7}}, 3});
    }
    if (n >= 10 && n <= 12) {
        // Single-pass sizes 2^10 .. 2^12, both word widths (BASELINE config 2 is 4-byte N = 2^12, batch 1024): below the batch that fills the SIMDs
        // the same unit runs on 512 threads x 8 words (variant 1: radix-8 rounds, one more LDS exchange) -- twice the waves per
        // polynomial, so a launch of one generation of workgroups issues its butterflies at 2 .. 6 waves per SIMD instead of
        // 1 .. 4 (VOP3 forms issue in 3.4 cycles per wave-instruction at 4 waves per SIMD, 2.0 at 8: profiles/r04_valu_issue_cost.json).
        // Same-process crossover (tools/ab_latency.py; round 4's driver script is in the history -> profiles/r04_ab_m32_wide.txt).  Alternative 0 = the wide variant,
        // alternative 1 = the default radix-16 kernel from the threshold on; same stages, same words.
        std::vector<PassDesc> wide = def;
        wide[0].variant = 1;
        alts.clear();
        alts.push_back({wide, 0});
        alts.push_back({def, word_bytes == 8 ? (uint64_t) NTT_ALT_MIN_BATCH_W8_WIDE
                             : p < 0x80000000ull ? (uint64_t) NTT_ALT_MIN_BATCH_M32_WIDE_LIGHT : (uint64_t) NTT_ALT_MIN_BATCH_M32_WIDE});
    }
    if (word_bytes == 8 && n == 13) alts.push_back({{{true, 0, 13}}, NTT_ALT_MIN_BATCH_GL13});
    if (word_bytes == 4 && m32_lazy_modulus(p) && n == 14) alts.push_back({{{true, 0, 14}}, NTT_ALT_MIN_BATCH_M32_14});
    return alts;
}

inline int select_alternative(const std::vector<PlanAlt> &alts, uint64_t batch) {
    int k = 0;
    for (size_t i = 1; i < alts.size(); i++)
        if (batch >= alts[i].min_batch) k = (int) i;
    return k;
}

// Montgomery constants of FieldM32
inline uint32_t mont_pinv(uint32_t p) {
    uint32_t inv = p;  // Newton: inv *= 2 - p*inv; 3 correct bits double each step
    for (int i = 0; i < 5; i++) inv *= 2u - p * inv;
    return inv;
}
inline uint32_t mont_r2(uint32_t p) { return (uint32_t) ((((u128) 1) << 64) % p); }
// ... and of FieldM64 (R = 2^64)
inline uint64_t mont_pinv64(uint64_t p) {
    uint64_t inv = p;
    for (int i = 0; i < 6; i++) inv *= 2ull - p * inv;
    return inv;
}
inline uint64_t mont_r2_64(uint64_t p) {
    const uint64_t r = (uint64_t) ((((u128) 1) << 64) % p);  // 2^64 mod p
    return mulmod(r, r, p);
}

// value -> the form the kernels keep twiddles in
inline uint64_t to_table_form(uint64_t t, uint64_t p, int word_bytes) {
    return (uint64_t) ((((u128) t) << (word_bytes == 4 ? 32 : 64)) % p);  // Montgomery form, R = 2^32 / 2^64
}

// Tinv[i] = T[i]^-1 for i >= 1 by batch inversion; false when some T[i] == 0
inline bool invert_table(const std::vector<uint64_t> &T, uint64_t p, std::vector<uint64_t> &Ti) {
    const size_t N = T.size();
    Ti.assign(N, 0);
    for (size_t i = 1; i < N; i++)
        if (T[i] == 0) return false;
    std::vector<uint64_t> pre(N);
    uint64_t acc = 1;
    for (size_t i = 1; i < N; i++) {
        pre[i] = acc;
        acc = mulmod(acc, T[i], p);
    }
    uint64_t inv = invmod(acc, p);
    if (inv == 0) return false;  // some entry shares a factor with p
    for (size_t i = N; i-- > 1;) {
        Ti[i] = mulmod(inv, pre[i], p);
        inv = mulmod(inv, T[i], p);
    }
    Ti[0] = 1;
    return true;
}

// table rules: kind 0 = src/test.cpp:27-32 + :138 (make_roots); 1 = cyclic bit-reversed;
// 2 = Longa-Naehrig psi^-1 bit-reversed.  false when N does not divide the order.
inline bool make_table(int kind, int logn, uint64_t p, uint64_t g, std::vector<uint64_t> &T) {
    const uint64_t N = 1ull << logn;
    T.assign(N, 0);
    if (kind == 0) {
        const uint64_t w = powmod(g, (p - 1) / N, p);
        T[0] = 1 % p;
        for (uint64_t i = 1; i < N; i++) T[i] = mulmod(T[i - 1], w, p);
        return true;
    }
    if (kind == 1) {
        if ((p - 1) % N) return false;
        const uint64_t w = powmod(g, (p - 1) / N, p);
        std::vector<uint64_t> pw(N);
        pw[0] = 1;
        for (uint64_t i = 1; i < N; i++) pw[i] = mulmod(pw[i - 1], w, p);
        T[0] = 1;
        int lh = 0;
        for (uint64_t h = 1; h < N; h <<= 1, lh++)
            for (uint64_t i = 0; i < h; i++) T[h + i] = pw[(bitrev(i, lh) * (N / (2 * h))) % N];
        return true;
    }
    if (kind == 2) {
        if ((p - 1) % (2 * N)) return false;
        const uint64_t psi = powmod(g, (p - 1) / (2 * N), p);
        const uint64_t psi_inv = invmod(psi, p);
        if (psi_inv == 0) return false;
        std::vector<uint64_t> pw(N);
        pw[0] = 1;
        for (uint64_t i = 1; i < N; i++) pw[i] = mulmod(pw[i - 1], psi_inv, p);
        for (uint64_t k = 0; k < N; k++) T[k] = pw[bitrev(k, logn)];
        return true;
    }
    return false;
}

}  // namespace host
}  // namespace ntt
