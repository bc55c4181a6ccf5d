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
//   FieldM64 : 8-byte words, ANY odd p < 2^64 (the reference's `%`-based network works for any modulus, src/test.cpp:48-50;
//              BASELINE's metric says "64-bit prime").  Montgomery with R = 2^64: eleven 32-bit multiplies per
//              product where the Goldilocks reduction needs four -- the general path, not the headline one.
//   FieldGL  : 8-byte words, p = 2^64 - 2^32 + 1.  Twiddles in Montgomery form
//              (T * 2^64 mod p); 64x64->128 product from four v_mad_u64_u32, then the
//              shift/add Montgomery reduction special to this prime.
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

struct FieldM64 {
    using W = uint64_t;
    uint64_t p;     // modulus (odd, < 2^64)
    uint64_t pinv;  // p^-1 mod 2^64
    uint64_t r2;    // 2^128 mod p (to enter Montgomery form)

    NTT_HD W add(W a, W b) const {
        const uint64_t s = a + b;  // may wrap when p > 2^63: keep the carry
        const bool carry = s < a;
        return (carry || s >= p) ? s - p : s;
    }
    NTT_HD W sub(W a, W b) const {
        const uint64_t d = a - b;
        return (a < b) ? d + p : d;
    }
    // x any 64-bit word, tw = T * 2^64 mod p (Montgomery form, < p)  ->  x*T mod p, canonical:
    //   t = x * tw < 2^64 * p;  m = lo(t) * p^-1 mod 2^64;  m*p has the same low half as t, so
    //   (t - m*p) / 2^64 = hi(t) - hi(m*p)  in (-p, p)
    NTT_HD W mul(W x, W tw) const {
        const unsigned __int128 t = (unsigned __int128) x * tw;
        const uint64_t lo = (uint64_t) t, hi = (uint64_t) (t >> 64);
        const uint64_t m = lo * pinv;
        const uint64_t mh = (uint64_t) (((unsigned __int128) m * p) >> 64);
        const uint64_t r = hi - mh;
        return (hi < mh) ? r + p : r;
    }
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
    // x canonical, tw = T * 2^64 mod p (Montgomery form)  ->  x*T mod p, canonical.
    // With p = 2^64 - 2^32 + 1, p^-1 = 1 + 2^32 (mod 2^64), so the Montgomery quotient
    // m = lo * p^-1 and (m*p) >> 64 are shifts and adds of the low product half:
    //   a = lo + (lo << 32)            (carry e)
    //   b = a - (a >> 32) - e          = (m*p) >> 64, always < p
    //   r = hi - b  (mod p)            one conditional  - (2^32 - 1)
    // (the shift/add form of Goldilocks Montgomery reduction used by plonky2).
    NTT_HD W mul(W x, W tw) const {
        const uint32_t x0 = (uint32_t) x, x1 = (uint32_t) (x >> 32);
        const uint32_t t0 = (uint32_t) tw, t1 = (uint32_t) (tw >> 32);
        const uint64_t ll = (uint64_t) x0 * t0;
        const uint64_t m1 = (uint64_t) x0 * t1 + (ll >> 32);   // fits: (2^32-1)^2 + 2^32-1 < 2^64
        const uint64_t q = (uint64_t) x1 * t0;
        const uint64_t m2 = q + m1;                            // may wrap
        const uint64_t cm = m2 < q ? 1u : 0u;
        const uint64_t hi = (uint64_t) x1 * t1 + ((cm << 32) | (m2 >> 32));
        const uint32_t p0 = (uint32_t) ll, p1 = (uint32_t) m2; // lo = p1:p0
        const uint32_t a1 = p1 + p0;
        const uint32_t e = a1 < p0 ? 1u : 0u;
        const uint64_t a = ((uint64_t) a1 << 32) | p0;
        const uint64_t b = a - a1 - e;
        const uint64_t r = hi - b;
        return (hi < b) ? r - EPS : r;
    }
    static constexpr uint64_t R2 = 0xFFFFFFFE00000001ULL;  // 2^128 mod p
    NTT_HD W mul_plain(W x, W y) const { return mul(mul(x, y), R2); }
    NTT_HD W to_table_form(W t) const { return mul(t, R2); }
};

}  // namespace ntt
