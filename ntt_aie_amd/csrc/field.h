// field.h -- modular arithmetic of the butterfly, written once for the gfx950
// device code and for the host-side index model (tests/emu).
//
// Replaces src/aie_core.cc:11-102 (modadd / modsub / barrett_2k and their
// 16-lane vector twins).  The reference reduces with Barrett (w = ceil(log2 p),
// u = floor(2^2w / p)); its results are canonical residues in [0, p), so any
// exact reduction gives the same words.  Here:
//   FieldM32 : 4-byte words, any odd p < 2^32.  Twiddles are kept in Montgomery
//              form (T * 2^32 mod p) so one product costs v_mad_u64_u32 +
//              v_mul_lo_u32 + v_mul_hi_u32 and returns the canonical x*T mod p.
//   FieldGL  : 8-byte words, p = 2^64 - 2^32 + 1.  64x64->128 product from four
//              v_mad_u64_u32, then the 2^64 = 2^32 - 1, 2^96 = -1 reduction.
// All inputs must be canonical; all outputs are canonical.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NTT_HD __host__ __device__ __forceinline__
#else
#define NTT_HD inline
#endif

namespace ntt {

struct FieldM32 {
    using W = uint32_t;
    uint32_t p;     // modulus (odd)
    uint32_t pinv;  // p^-1 mod 2^32
    uint32_t r2;    // 2^64 mod p (to enter Montgomery form)

    NTT_HD W add(W a, W b) const {
        // a + b can exceed 2^32 when p > 2^31: keep the carry.
        uint32_t s = a + b;
        bool carry = s < a;
        uint32_t t = s - p;
        return (carry || s >= p) ? t : s;
    }
    NTT_HD W sub(W a, W b) const {
        uint32_t d = a - b;
        return (a < b) ? d + p : d;
    }
    // x canonical, tw = T*2^32 mod p  ->  x*T mod p, canonical
    NTT_HD W mul(W x, W tw) const {
        uint64_t t = (uint64_t) x * tw;
        uint32_t lo = (uint32_t) t, hi = (uint32_t) (t >> 32);
        uint32_t m = lo * pinv;
        uint32_t mh = (uint32_t) (((uint64_t) m * p) >> 32);
        uint32_t r = hi - mh;  // (t - m*p) / 2^32, in (-p, p)
        return (hi < mh) ? r + p : r;
    }
    // plain x*y mod p for two normal-form operands
    NTT_HD W mul_plain(W x, W y) const { return mul(mul(x, y), r2); }
    NTT_HD W to_table_form(W t) const { return mul(t, r2); }
};

struct FieldGL {
    using W = uint64_t;
    static constexpr uint64_t P = 0xFFFFFFFF00000001ULL;
    static constexpr uint64_t EPS = 0xFFFFFFFFULL;  // 2^64 mod p

    NTT_HD W add(W a, W b) const {
        uint64_t s = a + b;
        bool c1 = s < a;          // true sum = s + 2^64; minus p = s + EPS (no 2nd wrap: sum < 2p)
        uint64_t t = s + EPS;
        bool c2 = t < s;          // s >= p
        return (c1 || c2) ? t : s;
    }
    NTT_HD W sub(W a, W b) const {
        uint64_t d = a - b;
        return (a < b) ? d - EPS : d;  // + p == - EPS (mod 2^64)
    }
    NTT_HD W mul(W a, W b) const {
        uint32_t a0 = (uint32_t) a, a1 = (uint32_t) (a >> 32);
        uint32_t b0 = (uint32_t) b, b1 = (uint32_t) (b >> 32);
        uint64_t p00 = (uint64_t) a0 * b0;
        uint64_t p01 = (uint64_t) a0 * b1 + (p00 >> 32);
        uint64_t p10 = (uint64_t) a1 * b0 + (uint32_t) p01;
        uint64_t hi = (uint64_t) a1 * b1 + (p01 >> 32) + (p10 >> 32);
        uint64_t lo = (p10 << 32) | (uint32_t) p00;
        return reduce128(lo, hi);
    }
    // lo + 2^64*hi mod p, canonical
    static NTT_HD W reduce128(uint64_t lo, uint64_t hi) {
        uint32_t hh = (uint32_t) (hi >> 32), hl = (uint32_t) hi;
        uint64_t t0 = lo - hh;                    // 2^96 = -1
        if (lo < hh) t0 -= EPS;
        uint64_t t1 = ((uint64_t) hl << 32) - hl; // hl * (2^32 - 1), 2^64 = 2^32 - 1
        uint64_t t2 = t0 + t1;
        if (t2 < t1) t2 += EPS;                   // cannot wrap twice: t1 <= 2^64 - 2^33 + 1
        uint64_t t3 = t2 + EPS;
        return (t3 < t2) ? t3 : t2;               // t2 >= p -> t2 - p
    }
    NTT_HD W mul_plain(W x, W y) const { return mul(x, y); }
    NTT_HD W to_table_form(W t) const { return t; }
};

}  // namespace ntt
