// asan_sweep.cpp -- the host index model under AddressSanitizer / UBSan, driven with EXACT-SIZE heap buffers.
//
// TEST INFRASTRUCTURE (tests/test_emu_asan.py builds and runs it; nothing in the product loads it).
// GPU sanitizers are not available on the pool, so the only detector for an out-of-bounds access of the pass kernels is
// the host index model: the same pass.h index rules compiled with g++.  numpy arrays and torch tensors hide over-reads
// (their allocators map the bytes behind a small array); here every coefficient buffer is a malloc() of exactly
// batch * N words and every table one of exactly N words, so ASan's red zones sit where a C caller's hipMalloc would end
// -- the reference host owns exactly N words per buffer too (src/test.cpp:115-124).
// Every case is also compared with the oracle (oracle/ntt_oracle.c, linked in), so a sweep that passes has run real work.
//
//   asan_sweep <family> [quick]     family: gl_fwd gl_inv m64_fwd m64_inv m32_fwd m32_inv prod_gl prod_m64 prod_m32
// built with -DEMU_PARTS=<the family's bits> (emu.cpp).  Exit code 0 = clean; 3 = wrong words; ASan / UBSan abort otherwise.
#include "emu.cpp"

#include <string>

extern "C" {
#include "../../oracle/ntt_oracle.h"
}

namespace {

struct Modulus {
    const char *name;
    int wb;
    uint64_t p, g;
    int max_logn;
};
const Modulus GL = {"goldilocks", 8, GOLDILOCKS, 7, 17};
const Modulus M64A = {"m64_62bit", 8, 0x3fffffee00000001ull, 0, 17};  // generator found below (any table with unit entries does)
const Modulus M64B = {"m64_above_2^63", 8, 0xfffffffc00000001ull, 0, 17};
const Modulus M32_LAZY = {"m32_lazy_998244353", 4, 998244353ull, 3, 17};
const Modulus M32_31 = {"m32_31bit_2013265921", 4, 2013265921ull, 31, 17};
const Modulus M32_32 = {"m32_32bit_3221225473", 4, 3221225473ull, 5, 17};
const Modulus M32_12 = {"m32_12bit_3329", 4, 3329ull, 3, 13};

uint64_t splitmix(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// exact-size heap words of either width
struct Buf {
    void *p = nullptr;
    int wb;
    size_t words;
    Buf(int wb_, size_t words_) : wb(wb_), words(words_) { p = malloc(words_ * (size_t) wb_); if (!p) abort(); }
    ~Buf() { free(p); }
    Buf(const Buf &) = delete;
    uint64_t get(size_t i) const { return wb == 4 ? ((const uint32_t *) p)[i] : ((const uint64_t *) p)[i]; }
    void set(size_t i, uint64_t v) { if (wb == 4) ((uint32_t *) p)[i] = (uint32_t) v; else ((uint64_t *) p)[i] = v; }
};

long g_cases = 0;

[[noreturn]] void fail(const std::string &what) {
    fprintf(stderr, "asan_sweep: WRONG WORDS in %s\n", what.c_str());
    exit(3);
}

// a table with unit entries for any modulus: kind 0 (make_roots, src/test.cpp:27-32); g = 0 means "3" (the entries only
// have to be invertible, which powers of anything coprime to p are)
void fill_table(const Modulus &m, int logn, Buf &T) {
    std::vector<uint64_t> t;
    make_table(0, logn, m.p, m.g ? m.g : 3, t);
    for (size_t i = 0; i < t.size(); i++) T.set(i, t[i]);
}

void oracle_forward(const Modulus &m, Buf &a, size_t N, size_t batch, const Buf &T) {
    if (m.wb == 4) oracle_ntt_batch_u32((uint32_t *) a.p, (uint32_t) N, batch, (const uint32_t *) T.p, (uint32_t) m.p, 1);
    else oracle_ntt_batch_u64((uint64_t *) a.p, N, batch, (const uint64_t *) T.p, m.p, 1);
}
void oracle_block(const Modulus &m, Buf &dst, const Buf &src, size_t N, size_t batch) {
    for (size_t b = 0; b < batch; b++) {
        if (m.wb == 4) oracle_block16_u32((uint32_t *) dst.p + b * N, (const uint32_t *) src.p + b * N, (uint32_t) N);
        else oracle_block16_u64((uint64_t *) dst.p + b * N, (const uint64_t *) src.p + b * N, N);
    }
}

std::string tag(const Modulus &m, int logn, size_t batch, int inverse, int layout, int scale, uint64_t ov, bool inplace) {
    char s[256];
    snprintf(s, sizeof s, "%s logN %d batch %zu %s layout %d scale %d passes 0x%llx variant %d%s", m.name, logn, batch,
             inverse ? "inverse" : "forward", layout, scale, (unsigned long long) (ov & ((1ull << 60) - 1)), (int) (ov >> 60),
             inplace ? " in place" : "");
    return s;
}

// one transform on exact-size buffers against the oracle
void one(const Modulus &m, int logn, size_t batch, int inverse, int layout, int scale, uint64_t ov, bool inplace, uint32_t target_wgs) {
    const size_t N = (size_t) 1 << logn, words = N * batch;
    Buf T(m.wb, N), x(m.wb, words), y(m.wb, words);
    fill_table(m, logn, T);
    uint64_t seed = 0x1234 + (uint64_t) logn * 131 + batch;
    for (size_t i = 0; i < words; i++) x.set(i, splitmix(seed) % m.p);
    memcpy(y.p, x.p, words * (size_t) m.wb);
    oracle_forward(m, y, N, batch, T);  // y = network(x)
    Buf src(m.wb, words), want(m.wb, words);
    if (!inverse) {
        memcpy(src.p, x.p, words * (size_t) m.wb);
        if (layout) oracle_block(m, want, y, N, batch);
        else memcpy(want.p, y.p, words * (size_t) m.wb);
    } else {
        if (layout) oracle_block(m, src, y, N, batch);
        else memcpy(src.p, y.p, words * (size_t) m.wb);
        const uint64_t nmod = (uint64_t) (((u128) 1 << logn) % m.p);
        for (size_t i = 0; i < words; i++) want.set(i, scale ? x.get(i) : mulmod(x.get(i), nmod, m.p));
    }
    const std::string t = tag(m, logn, batch, inverse, layout, scale, ov, inplace);
    int rc;
    if (inplace) {
        rc = emu_transform(m.wb, logn, m.p, T.p, src.p, src.p, (uint32_t) batch, inverse, layout, scale, target_wgs, ov);
        if (rc == 0 && memcmp(src.p, want.p, words * (size_t) m.wb)) fail(t);
    } else {
        Buf out(m.wb, words);
        memset(out.p, 0xEE, words * (size_t) m.wb);
        rc = emu_transform(m.wb, logn, m.p, T.p, src.p, out.p, (uint32_t) batch, inverse, layout, scale, target_wgs, ov);
        if (rc == 0 && memcmp(out.p, want.p, words * (size_t) m.wb)) fail(t);
    }
    if (rc) {
        fprintf(stderr, "asan_sweep: emu_transform rc %d in %s\n", rc, t.c_str());
        exit(4);
    }
    ++g_cases;
}

uint64_t pack(const std::vector<int> &ms, int variant) {
    uint64_t v = 0;
    for (size_t i = 0; i < ms.size(); i++) v |= (uint64_t) ms[i] << (4 * i);
    return v | ((uint64_t) variant << 60);
}

// every decomposition the kernels exist for: the planner's default, every plan alternative with its kernel variants, and
// every (first, column...) split of the tile shapes (CONTIG 1..13, 14 for 4-byte words; column 4..9; first pass wide
// enough for a column tile: 16 words of 8 bytes, 32 of 4)
std::vector<uint64_t> decompositions(const Modulus &m, int logn, bool quick) {
    std::vector<uint64_t> out;
    out.push_back(0);  // plan_passes()
    const auto alts = plan_alternatives(logn, m.wb, m.p);
    for (const auto &alt : alts) {
        std::vector<int> ms;
        for (const auto &ps : alt.passes) ms.push_back(ps.log_m);
        out.push_back(pack(ms, alt.passes[0].variant));
    }
    if (logn >= 10 && logn <= 12) out.push_back(pack({logn}, 1));
    const int top = m.wb == 4 ? 14 : 13, min_first = m.wb == 4 ? 5 : 4;
    for (int m1 = 4; m1 <= 9; m1++) {
        const int m0 = logn - m1;
        if (m0 >= min_first && m0 <= top && !(quick && (m1 % 2))) out.push_back(pack({m0, m1}, 0));
    }
    if (logn >= 13 && !quick)
        for (int m1 = 4; m1 <= 5; m1++)
            for (int m2 = 4; m2 <= 9; m2 += 5) {
                const int m0 = logn - m1 - m2;
                if (m0 >= min_first && m0 <= 9) out.push_back(pack({m0, m1, m2}, 0));
            }
    // no duplicates
    std::vector<uint64_t> uniq;
    for (uint64_t v : out) {
        bool seen = false;
        for (uint64_t u : uniq) seen |= u == v;
        if (!seen) uniq.push_back(v);
    }
    return uniq;
}

std::vector<size_t> batches(int logn, bool quick) {
    // the small odd ones, and 2^k + 1: ragged against every number of polynomials a workgroup can hold (N = 2: 2048 of them)
    const size_t budget = (size_t) 1 << (quick ? 15 : 18);  // words per case
    std::vector<size_t> ok;
    for (size_t v : {1, 3, 5, 9, 17, 65, 257, 2049})
        if (v == 1 || (v << logn) <= budget || (v <= 5 && logn <= 14) || (v == 3 && logn <= 16)) ok.push_back(v);
    return ok;
}

void sweep_transform(const std::vector<Modulus> &mods, int inverse, bool quick) {
    for (const Modulus &m : mods)
        for (int logn = 1; logn <= m.max_logn; logn++) {
            const auto decs = decompositions(m, logn, quick);
            const auto bs = batches(logn, quick);
            for (size_t di = 0; di < decs.size(); di++)
                for (size_t bi = 0; bi < bs.size(); bi++) {
                    const size_t batch = bs[bi];
                    // both layouts (AIE_BLOCK16 needs N >= 16), the inverse scaled and unscaled; in place / out of place and the
                    // launch target alternate (a small target: several polynomial groups per workgroup and tapered rows even
                    // at these batches).  quick: one layout / scaling per case, alternating.
                    int k = 0;
                    for (int layout = 0; layout <= (logn >= 4 ? 1 : 0); layout++)
                        for (int scale = 1; scale >= (inverse ? 0 : 1); scale--, k++) {
                            if (quick && ((int) (bi + di) & 1) != layout) continue;
                            if (quick && inverse && ((int) (bi + di / 2) & 1) != scale) continue;
                            one(m, logn, batch, inverse, layout, scale, decs[di], ((bi + di + (size_t) k) % 3) == 0, ((bi + (size_t) k) & 1) ? 4 : 2048);
                        }
                }
        }
}

// forward of (a . b . scale): the fused first pass (emu_forward_product), and the fused negacyclic product (emu_polymul_fused)
void sweep_product(const Modulus &m, bool quick) {
    for (int logn = 1; logn <= m.max_logn; logn++) {
        const size_t N = (size_t) 1 << logn;
        for (size_t batch : batches(logn, true)) {
            if (quick && batch > 9) continue;
            const size_t words = N * batch;
            uint64_t seed = 77 + (uint64_t) logn * 7 + batch;
            if (m.p == GOLDILOCKS || m.wb == 4) {  // emu_forward_product dispatches these two families
                Buf T(m.wb, N), a(m.wb, words), b(m.wb, words), out(m.wb, words), want(m.wb, words);
                fill_table(m, logn, T);
                for (size_t i = 0; i < words; i++) {
                    a.set(i, splitmix(seed) % m.p);
                    b.set(i, splitmix(seed) % m.p);
                }
                const uint64_t sc = splitmix(seed) % m.p;
                for (size_t i = 0; i < words; i++) want.set(i, mulmod(mulmod(a.get(i), b.get(i), m.p), sc, m.p));
                oracle_forward(m, want, N, batch, T);
                const int rc = emu_forward_product(m.wb, logn, m.p, T.p, a.p, b.p, out.p, (uint32_t) batch, sc, (batch & 2) ? 4 : 2048);
                char s[128];
                snprintf(s, sizeof s, "forward_product %s logN %d batch %zu", m.name, logn, batch);
                if (rc) { fprintf(stderr, "asan_sweep: rc %d in %s\n", rc, s); exit(4); }
                if (memcmp(out.p, want.p, words * (size_t) m.wb)) fail(s);
                ++g_cases;
            }
            // negacyclic product with the kind-2 table, where the modulus has a 2N-th root of unity and the first pass a product kernel
            std::vector<uint64_t> t2;
            uint64_t g = 2;  // a quadratic non-residue: g^((p-1)/2N) then has order exactly 2N (the sweep's moduli are prime)
            while (powmod(g, (m.p - 1) / 2, m.p) != m.p - 1) ++g;
            if (!make_table(2, logn, m.p, g, t2)) continue;
            const int m0 = plan_passes(logn, m.wb)[0].log_m;
            if (m.wb == 8 ? (m0 < 7 || m0 > 12) : (m0 < 5 || m0 > 13)) continue;
            Buf T(m.wb, N), a(m.wb, words), b(m.wb, words), out(m.wb, words);
            for (size_t i = 0; i < N; i++) T.set(i, t2[i]);
            for (size_t i = 0; i < words; i++) {
                a.set(i, splitmix(seed) % m.p);
                b.set(i, splitmix(seed) % m.p);
            }
            // expected: c = Fwd( InvU(a) . InvU(b) . N^-1 ) = Fwd( Inv(a) . Inv(b) . N ), Inv = the oracle's exact inverse
            Buf ia(m.wb, words), ib(m.wb, words), want(m.wb, words);
            memcpy(ia.p, a.p, words * (size_t) m.wb);
            memcpy(ib.p, b.p, words * (size_t) m.wb);
            int orc;
            if (m.wb == 4) orc = oracle_intt_batch_u32((uint32_t *) ia.p, (uint32_t) N, batch, (const uint32_t *) T.p, (uint32_t) m.p, 1) |
                                 oracle_intt_batch_u32((uint32_t *) ib.p, (uint32_t) N, batch, (const uint32_t *) T.p, (uint32_t) m.p, 1);
            else orc = oracle_intt_batch_u64((uint64_t *) ia.p, N, batch, (const uint64_t *) T.p, m.p, 1) |
                       oracle_intt_batch_u64((uint64_t *) ib.p, N, batch, (const uint64_t *) T.p, m.p, 1);
            if (orc) abort();
            const uint64_t nmod = (uint64_t) (((u128) 1 << logn) % m.p);
            for (size_t i = 0; i < words; i++) want.set(i, mulmod(mulmod(ia.get(i), ib.get(i), m.p), nmod, m.p));
            oracle_forward(m, want, N, batch, T);
            const int rc = emu_polymul_fused(m.wb, logn, m.p, T.p, a.p, b.p, out.p, (uint32_t) batch, (batch & 2) ? 4 : 2048);
            char s[128];
            snprintf(s, sizeof s, "polymul_fused %s logN %d batch %zu", m.name, logn, batch);
            if (rc) { fprintf(stderr, "asan_sweep: rc %d in %s\n", rc, s); exit(4); }
            if (memcmp(out.p, want.p, words * (size_t) m.wb)) fail(s);
            ++g_cases;
        }
    }
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: asan_sweep <family> [quick]\n");
        return 2;
    }
    const std::string fam = argv[1];
    const bool quick = argc > 2 && std::string(argv[2]) == "quick";
    emu_set_tracking(0);
    const std::vector<Modulus> m32 = {M32_LAZY, M32_31, M32_32, M32_12};
    if (fam == "gl_fwd") sweep_transform({GL}, 0, quick);
    else if (fam == "gl_inv") sweep_transform({GL}, 1, quick);
    else if (fam == "m64_fwd") sweep_transform({M64B, M64A}, 0, quick);
    else if (fam == "m64_inv") sweep_transform({M64B, M64A}, 1, quick);
    else if (fam == "m32_fwd") sweep_transform(m32, 0, quick);
    else if (fam == "m32_inv") sweep_transform(m32, 1, quick);
    else if (fam == "prod_gl") sweep_product(GL, quick);
    else if (fam == "prod_m64") sweep_product(M64B, quick);
    else if (fam == "prod_m32") { sweep_product(M32_32, quick); sweep_product(M32_LAZY, quick); }
    else return 2;
    printf("asan_sweep %s: %ld cases clean\n", fam.c_str(), g_cases);
    return 0;
}
