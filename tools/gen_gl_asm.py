#!/usr/bin/env python3
"""Generate ntt_aie_amd/csrc/gl_asm.h: hand-scheduled gfx950 instruction streams for the
Goldilocks butterflies (two independent butterflies interleaved per asm statement).

Why asm: on gfx950 a VALU instruction that reads an SGPR/VCC written by a VALU instruction
needs two other instructions in between (the compiler pads with s_nop), and hipcc turns the
64-bit compare/select idiom into v_cmp_lt_u64 + 2x v_cndmask instead of carry chains.  One
butterfly is 22 VALU instructions when written with v_add_co/v_addc_co/v_subb_co chains:

  sub   d = x - y (mod p)                    4 VALU + 1 SALU
  mul   r = d * T  (T in Montgomery form)    4 v_mad_u64_u32 + 9 VALU + 2 SALU (zero half of one addend pinned; merged reduction)
  add   s = x + y (mod p)                    5 VALU + 2 SALU

The two butterflies' instructions are merged by a list scheduler that keeps every
SGPR producer->consumer pair at least 3 slots apart, so no s_nop is needed.

Register use inside a statement: data words and 32-bit temporaries are compiler-allocated
operands; the 64-bit products need aligned VGPR pairs whose halves are used separately, which
inline-asm operands cannot express, so they live in fixed VGPRs v[108:127] (clobbers), and the
carries in fixed SGPR pairs s[80:99].  The s_or_b64 / s_andn2_b64 in the streams write
SCC, so SCC is declared clobbered too (without it hipcc scheduled an s_add_u32 / s_addc_u32 pair
across a statement and the carry was lost).
"""
import sys

import os

SGPR_DIST = 3  # producer slot + 3 <= consumer slot  (two instructions in between)
M32_VBASE = int(os.environ.get("M32_VBASE", "72"))  # first of the 8 fixed VGPRs of the 4-byte-word streams


class Ins:
    def __init__(self, text, reads=(), writes=(), salu=False):
        self.text, self.reads, self.writes, self.salu = text, set(reads), set(writes), salu


def is_sgpr(r):
    return r.startswith("s") or r == "vcc"


def butterfly(kind, b, vbase=104, sbase=80):
    """Instruction list of butterfly `b` (0/1).  kind: 'fwd' | 'inv' | 'mul'.
    Operand names: x0 x1 y0 y1 t0 t1 (compiler operands, suffixed by b), temporaries
    d0 d1 (operands), fixed pairs L M A B H Z, SGPR pairs sa sb se sf."""
    vb = vbase + 12 * b
    L, M, A, B, H, Z = [(f"v{vb + 2 * i}", f"v{vb + 2 * i + 1}") for i in range(6)]
    sb_ = sbase + 10 * b
    sa, sbb, se, sf, s5 = [f"s[{sb_ + 2 * i}:{sb_ + 2 * i + 1}]" for i in range(5)]  # s5: the product's sign while its fix-up is computed

    def P(pair):
        lo = int(pair[0][1:])
        return f"v[{lo}:{lo + 1}]"

    def o(name):  # compiler-allocated operand
        return f"%[{name}{b}]"

    x0, x1, y0, y1, t0, t1 = o("x0_"), o("x1_"), o("y0_"), o("y1_"), o("t0_"), o("t1_")
    d0, d1 = o("d0_"), o("d1_")
    ins = []

    def sub(dst0, dst1, a0, a1, b0, b1, tmp0, tmp1, final0, final1):
        # (a - b) mod p: wrapped difference, then - borrow*(2^32-1)
        ins.append(Ins(f"v_sub_co_u32 {tmp0}, {sa}, {a0}, {b0}", [a0, b0], [tmp0, sa]))
        ins.append(Ins(f"v_subb_co_u32 {tmp1}, {sa}, {a1}, {b1}, {sa}", [a1, b1, sa], [tmp1, sa]))
        ins.append(Ins(f"v_addc_co_u32 {final0}, {sbb}, {tmp0}, 0, {sa}", [tmp0, sa], [final0, sbb]))
        ins.append(Ins(f"s_andn2_b64 {sa}, {sa}, {sbb}", [sa, sbb], [sa], salu=True))
        ins.append(Ins(f"v_subbrev_co_u32 {final1}, {sa}, 0, {tmp1}, {sa}", [tmp1, sa], [final1, sa]))

    def add(a0, a1, b0, b1, out0, out1):
        # (a + b) mod p, canonical: z = a + b (carry c); e = c | (z >= p) by ONE 64-bit compare; out = z + e*(2^32-1)
        # as out0 = z0 - e (borrow g), out1 = z1 + (e & ~g).  5 VALU + 2 SALU (was 6 + 1 with two selects).
        ins.append(Ins(f"v_add_co_u32 {Z[0]}, {se}, {a0}, {b0}", [a0, b0], [Z[0], se]))
        ins.append(Ins(f"v_addc_co_u32 {Z[1]}, {se}, {a1}, {b1}, {se}", [a1, b1, se], [Z[1], se]))
        ins.append(Ins(f"v_cmp_le_u64 {sf}, %[pp], {P(Z)}", [Z[0], Z[1]], [sf]))
        ins.append(Ins(f"s_or_b64 {se}, {se}, {sf}", [se, sf], [se], salu=True))
        ins.append(Ins(f"v_subbrev_co_u32 {out0}, {sf}, 0, {Z[0]}, {se}", [Z[0], se], [out0, sf]))
        ins.append(Ins(f"s_andn2_b64 {se}, {se}, {sf}", [se, sf], [se], salu=True))
        ins.append(Ins(f"v_addc_co_u32 {out1}, {sf}, {Z[1]}, 0, {se}", [Z[1], se], [out1, sf]))

    def lazy_add(a0, a1, b0, b1):
        # a += b (mod p) where b is canonical and a is ANY 64-bit representative; the result is again any
        # representative: sum, then + carry*(2^32-1), which cannot carry twice because b < p.  4 VALU + 1 SALU.
        # (Inverse / DIT butterflies only: there one operand of every sum is a fresh product.)
        ins.append(Ins(f"v_add_co_u32 {a0}, {se}, {a0}, {b0}", [a0, b0], [a0, se]))
        ins.append(Ins(f"v_addc_co_u32 {a1}, {se}, {a1}, {b1}, {se}", [a1, b1, se], [a1, se]))
        ins.append(Ins(f"v_subbrev_co_u32 {a0}, {sf}, 0, {a0}, {se}", [a0, se], [a0, sf]))
        ins.append(Ins(f"s_andn2_b64 {se}, {se}, {sf}", [se, sf], [se], salu=True))
        ins.append(Ins(f"v_addc_co_u32 {a1}, {sf}, {a1}, 0, {se}", [a1, se], [a1, sf]))

    def mul(m0, m1, r0, r1, t0=t0, t1=t1):
        # (m1:m0) * (t1:t0) * 2^-64 mod p, canonical; m may be any 64-bit value.
        # Product P = (P3:P2:P1:P0) by four v_mad_u64_u32 (the carry-outs that mean nothing go to sbb), H = (P3:P2).
        ins.append(Ins(f"v_mad_u64_u32 {P(L)}, {sbb}, {m0}, {t0}, 0", [m0, t0], [L[0], L[1], sbb]))
        ins.append(Ins(f"v_mov_b32 {A[0]}, {L[1]}", [L[1]], [A[0]]))
        # A[1] holds zero for the whole kernel: a register-pinned variable handed to the statement as an INPUT (emit()),
        # so the compiler materialises it once outside the batch loop instead of one v_mov per butterfly
        ins.append(Ins(f"v_mad_u64_u32 {P(M)}, {sbb}, {m0}, {t1}, {P(A)}", [m0, t1, A[0], A[1]], [M[0], M[1], sbb]))
        ins.append(Ins(f"v_mad_u64_u32 {P(M)}, {sa}, {m1}, {t0}, {P(M)}", [m1, t0, M[0], M[1]], [M[0], M[1], sa]))
        ins.append(Ins(f"v_mov_b32 {B[0]}, {M[1]}", [M[1]], [B[0]]))
        ins.append(Ins(f"v_cndmask_b32 {B[1]}, 0, 1, {sa}", [sa], [B[1]]))
        ins.append(Ins(f"v_mad_u64_u32 {P(H)}, {sbb}, {m1}, {t1}, {P(B)}", [m1, t1, B[0], B[1]], [H[0], H[1], sbb]))
        # Montgomery step for p = 2^64 - 2^32 + 1 (p^-1 = 2^32 + 1 mod 2^64): q = (a1:P0) with a1 = P1 + P0 (carry e), and
        #   (P - q*p) / 2^64 = H - q + a1 + e = (H1 - a1 - e) * 2^32 + (H0 + P1 + e)        in (-p, p)
        # (because a1 - P0 = P1 - e * 2^32): one add for a1, one add-with-carry per half, + p when negative.  6 VALU + 2 SALU
        # (the q - a1 - e / H - b form it replaces was 7 + 1).
        ins.append(Ins(f"v_add_co_u32 {A[0]}, {sa}, {M[0]}, {L[0]}", [M[0], L[0]], [A[0], sa]))                      # a1, e
        ins.append(Ins(f"v_subb_co_u32 {L[1]}, {s5}, {H[1]}, {A[0]}, {sa}", [H[1], A[0], sa], [L[1], s5]))             # t = H1 - a1 - e, borrow
        ins.append(Ins(f"v_addc_co_u32 {L[0]}, {sbb}, {H[0]}, {M[0]}, {sa}", [H[0], M[0], sa], [L[0], sbb]))           # lo = H0 + P1 + e, carry c
        ins.append(Ins(f"v_addc_co_u32 {L[1]}, {sa}, {L[1]}, 0, {sbb}", [L[1], sbb], [L[1], sa]))                      # hi = t + c, carry g
        ins.append(Ins(f"s_andn2_b64 {s5}, {s5}, {sa}", [s5, sa], [s5], salu=True))                                    # negative = borrow & ~g
        ins.append(Ins(f"v_addc_co_u32 {r0}, {sbb}, {L[0]}, 0, {s5}", [L[0], s5], [r0, sbb]))                          # + p = - (2^32 - 1): lo + 1 ...
        ins.append(Ins(f"s_andn2_b64 {s5}, {s5}, {sbb}", [s5, sbb], [s5], salu=True))
        ins.append(Ins(f"v_subbrev_co_u32 {r1}, {s5}, 0, {L[1]}, {s5}", [L[1], s5], [r1, s5]))                         # ... hi - 1 unless lo carried

    if kind == "fwd":    # x' = x + y ; y' = (x - y) * T
        sub(d0, d1, x0, x1, y0, y1, d0, d1, d0, d1)
        add(x0, x1, y0, y1, x0, x1)
        mul(d0, d1, y0, y1)
    elif kind == "inv":  # w = y * T ; x' = x + w ; y' = x - w
        mul(y0, y1, d0, d1)
        sub(y0, y1, x0, x1, d0, d1, y0, y1, y0, y1)   # x may be any representative, w = d is canonical
        lazy_add(x0, x1, d0, d1)
    elif kind == "mul":  # x' = x * T
        mul(x0, x1, x0, x1)
    elif kind == "invs":  # LAST inverse stage with the N^-1 scaling folded in: w = y * (T*c) ; u = x * c ; x' = u + w ; y' = u - w
        # t = T^-1 * N^-1 (a plan-time table, per block), c = N^-1 (wave-uniform, SGPRs): N/2 extra products per transform
        # instead of the N of a separate scaling sweep.  x may be any 64-bit representative; both outputs are canonical.
        mul(y0, y1, d0, d1)
        mul(x0, x1, x0, x1, "%[c0]", "%[c1]")
        sub(y0, y1, x0, x1, d0, d1, y0, y1, y0, y1)   # u, w canonical -> canonical difference
        add(x0, x1, d0, d1, x0, x1)
    return ins


def butterfly64(kind, b, vbase=104):
    """General odd 64-bit modulus (FieldM64, Montgomery R = 2^64): instruction list of butterfly `b`.  kind 'fwd64' | 'inv64' | 'mul64'.
    23 VALU per product: 128-bit product (4 v_mad_u64_u32 + 3 glue), q = lo * p^-1 mod 2^64 (1 mad + 2 v_mul_lo + 1 add3),
    hi64(q * p) (v_mul_hi + 3 mads + 2 glue; its low half equals the product's, so only the carry chain is computed),
    r = hi - hi64(q*p), + p when negative (6).  Canonical modadd / modsub 6 each: 35 VALU per butterfly where hipcc's code for
    the portable FieldM64 takes 42.6 + 4.8 s_nop.  Scalars: %[p0] %[p1] %[pi0] %[pi1] (SGPRs: one per instruction, the gfx9
    constant-bus limit); the corrections that combine p with a carry held in an SGPR read p from two register-pinned VGPRs
    %[vp0] %[vp1].  Fixed VGPR pairs as in butterfly(): L M A B H Z, A[1] pinned to zero."""
    vb = vbase + 12 * b
    L, M, A, B, H, Z = [(f"v{vb + 2 * i}", f"v{vb + 2 * i + 1}") for i in range(6)]
    sb_ = 80 + 10 * b
    sa, sbb, se, sf, s5 = [f"s[{sb_ + 2 * i}:{sb_ + 2 * i + 1}]" for i in range(5)]

    def P(pair):
        lo = int(pair[0][1:])
        return f"v[{lo}:{lo + 1}]"

    def o(name):
        return f"%[{name}{b}]"

    x0, x1, y0, y1, t0, t1 = o("x0_"), o("x1_"), o("y0_"), o("y1_"), o("t0_"), o("t1_")
    d0, d1 = o("d0_"), o("d1_")
    P0, P1, PI0, PI1, VP0, VP1 = "%[p0]", "%[p1]", "%[pi0]", "%[pi1]", "%[vp0]", "%[vp1]"
    ins = []

    def sub(a0, a1, b0, b1, out0, out1):
        # (a - b) mod p, canonical operands: wrapped difference, + p when it borrowed.  6 VALU
        ins.append(Ins(f"v_sub_co_u32 {Z[0]}, {se}, {a0}, {b0}", [a0, b0], [Z[0], se]))
        ins.append(Ins(f"v_subb_co_u32 {Z[1]}, {se}, {a1}, {b1}, {se}", [a1, b1, se], [Z[1], se]))
        ins.append(Ins(f"v_add_co_u32 {B[0]}, {sf}, {Z[0]}, {VP0}", [Z[0], VP0], [B[0], sf]))
        ins.append(Ins(f"v_addc_co_u32 {B[1]}, {sf}, {Z[1]}, {VP1}, {sf}", [Z[1], VP1, sf], [B[1], sf]))
        ins.append(Ins(f"v_cndmask_b32 {out0}, {Z[0]}, {B[0]}, {se}", [Z[0], B[0], se], [out0]))
        ins.append(Ins(f"v_cndmask_b32 {out1}, {Z[1]}, {B[1]}, {se}", [Z[1], B[1], se], [out1]))

    def add(a0, a1, b0, b1, out0, out1):
        # (a + b) mod p, canonical operands: z = a + b (carry c), w = z - p (borrow g); out = (g & ~c) ? z : w.  6 VALU + 1 SALU
        ins.append(Ins(f"v_add_co_u32 {Z[0]}, {se}, {a0}, {b0}", [a0, b0], [Z[0], se]))
        ins.append(Ins(f"v_addc_co_u32 {Z[1]}, {se}, {a1}, {b1}, {se}", [a1, b1, se], [Z[1], se]))
        ins.append(Ins(f"v_sub_co_u32 {B[0]}, {sf}, {Z[0]}, {VP0}", [Z[0], VP0], [B[0], sf]))
        ins.append(Ins(f"v_subb_co_u32 {B[1]}, {sf}, {Z[1]}, {VP1}, {sf}", [Z[1], VP1, sf], [B[1], sf]))
        ins.append(Ins(f"s_andn2_b64 {sf}, {sf}, {se}", [sf, se], [sf], salu=True))
        ins.append(Ins(f"v_cndmask_b32 {out0}, {B[0]}, {Z[0]}, {sf}", [B[0], Z[0], sf], [out0]))
        ins.append(Ins(f"v_cndmask_b32 {out1}, {B[1]}, {Z[1]}, {sf}", [B[1], Z[1], sf], [out1]))

    def mul(m0, m1, r0, r1, t0=t0, t1=t1):
        # (m1:m0) * (t1:t0) * 2^-64 mod p, canonical; m may be any 64-bit value, t < p
        ins.append(Ins(f"v_mad_u64_u32 {P(L)}, {sbb}, {m0}, {t0}, 0", [m0, t0], [L[0], L[1], sbb]))
        ins.append(Ins(f"v_mov_b32 {A[0]}, {L[1]}", [L[1]], [A[0]]))
        ins.append(Ins(f"v_mad_u64_u32 {P(M)}, {sbb}, {m0}, {t1}, {P(A)}", [m0, t1, A[0], A[1]], [M[0], M[1], sbb]))
        ins.append(Ins(f"v_mad_u64_u32 {P(M)}, {sa}, {m1}, {t0}, {P(M)}", [m1, t0, M[0], M[1]], [M[0], M[1], sa]))
        ins.append(Ins(f"v_mov_b32 {B[0]}, {M[1]}", [M[1]], [B[0]]))
        ins.append(Ins(f"v_cndmask_b32 {B[1]}, 0, 1, {sa}", [sa], [B[1]]))
        ins.append(Ins(f"v_mad_u64_u32 {P(H)}, {sbb}, {m1}, {t1}, {P(B)}", [m1, t1, B[0], B[1]], [H[0], H[1], sbb]))   # H = P3:P2 ; lo = M0:L0
        # q = lo * p^-1 mod 2^64
        ins.append(Ins(f"v_mad_u64_u32 {P(Z)}, {sbb}, {L[0]}, {PI0}, 0", [L[0]], [Z[0], Z[1], sbb]))
        ins.append(Ins(f"v_mul_lo_u32 {B[0]}, {L[0]}, {PI1}", [L[0]], [B[0]]))
        ins.append(Ins(f"v_mul_lo_u32 {B[1]}, {M[0]}, {PI0}", [M[0]], [B[1]]))
        ins.append(Ins(f"v_add3_u32 {Z[1]}, {Z[1]}, {B[0]}, {B[1]}", [Z[1], B[0], B[1]], [Z[1]]))                              # q = Z1:Z0
        # hi64(q * p): (q*p) mod 2^64 == lo by construction, so only the carries into the high half are needed
        ins.append(Ins(f"v_mul_hi_u32 {A[0]}, {Z[0]}, {P0}", [Z[0]], [A[0]]))                                                  # A = {hi32(q0 p0), 0}
        ins.append(Ins(f"v_mad_u64_u32 {P(L)}, {sbb}, {Z[0]}, {P1}, {P(A)}", [Z[0], A[0], A[1]], [L[0], L[1], sbb]))
        ins.append(Ins(f"v_mad_u64_u32 {P(L)}, {sa}, {Z[1]}, {P0}, {P(L)}", [Z[1], L[0], L[1]], [L[0], L[1], sa]))
        ins.append(Ins(f"v_mov_b32 {B[0]}, {L[1]}", [L[1]], [B[0]]))
        ins.append(Ins(f"v_cndmask_b32 {B[1]}, 0, 1, {sa}", [sa], [B[1]]))
        ins.append(Ins(f"v_mad_u64_u32 {P(M)}, {sbb}, {Z[1]}, {P1}, {P(B)}", [Z[1], B[0], B[1]], [M[0], M[1], sbb]))           # M = hi64(q p)
        # r = H - M, + p when it borrowed
        ins.append(Ins(f"v_sub_co_u32 {H[0]}, {s5}, {H[0]}, {M[0]}", [H[0], M[0]], [H[0], s5]))
        ins.append(Ins(f"v_subb_co_u32 {H[1]}, {s5}, {H[1]}, {M[1]}, {s5}", [H[1], M[1], s5], [H[1], s5]))
        ins.append(Ins(f"v_add_co_u32 {Z[0]}, {sa}, {H[0]}, {VP0}", [H[0], VP0], [Z[0], sa]))
        ins.append(Ins(f"v_addc_co_u32 {Z[1]}, {sa}, {H[1]}, {VP1}, {sa}", [H[1], VP1, sa], [Z[1], sa]))
        ins.append(Ins(f"v_cndmask_b32 {r0}, {H[0]}, {Z[0]}, {s5}", [H[0], Z[0], s5], [r0]))
        ins.append(Ins(f"v_cndmask_b32 {r1}, {H[1]}, {Z[1]}, {s5}", [H[1], Z[1], s5], [r1]))

    if kind == "fwd64":    # x' = x + y ; y' = (x - y) * T
        sub(x0, x1, y0, y1, d0, d1)
        add(x0, x1, y0, y1, x0, x1)
        mul(d0, d1, y0, y1)
    elif kind == "inv64":  # w = y * T ; x' = x + w ; y' = x - w   (all canonical)
        mul(y0, y1, d0, d1)
        sub(x0, x1, d0, d1, y0, y1)
        add(x0, x1, d0, d1, x0, x1)
    elif kind == "mul64":  # x' = x * T
        mul(x0, x1, x0, x1)
    elif kind == "invs64":  # last inverse stage with N^-1 folded in (see butterfly(): "invs"): t = T^-1 * N^-1, c = N^-1 in SGPRs
        mul(y0, y1, d0, d1)
        mul(x0, x1, x0, x1, "%[c0]", "%[c1]")
        sub(x0, x1, d0, d1, y0, y1)
        add(x0, x1, d0, d1, x0, x1)
    return ins


def butterfly32(kind, b, mode, vbase=None):
    """4-byte-word Montgomery butterfly `b` (R = 2^32, twiddle in Montgomery form).
    mode "lazy":  p < 2^30, values kept in [0, 2p) (Harvey): no correction after the product, one
                  v_min_u32 per sum; 9 (forward, one of them a plain move) / 9 (inverse) instructions.  Callers
                  canonicalise once at the end of the transform.
    mode "small": p < 2^31, canonical values, conditional corrections by v_min_u32 (11 instructions);
    mode "any":   any odd p < 2^32, carries in SGPR pairs (12 instructions).
    The Montgomery step is a second multiply-add (m*T + u*p, u = lo(m*T) * (-p^-1)): 3 instructions for the lazy product.
    Operands: x y t (compiler), temporaries a b c, scalars %[p] %[npinv] = -p^-1 mod 2^32 (%[p2] = 2p when lazy); the
    64-bit product lives in a fixed VGPR pair."""
    small = mode == "small"
    L = (f"v{vbase + 2 * b}", f"v{vbase + 2 * b + 1}")
    LP = f"v[{vbase + 2 * b}:{vbase + 2 * b + 1}]"
    sa, sb = f"s[{84 + 4 * b}:{85 + 4 * b}]", f"s[{86 + 4 * b}:{87 + 4 * b}]"

    def o(name):
        return f"%[{name}{b}]"

    x, y, t, ta, tb, tc = o("x_"), o("y_"), o("t_"), o("a_"), o("b_"), o("c_")
    P, NPI, P2 = "%[p]", "%[npinv]", "%[p2]"
    ins = []

    def add(u, v, out):  # out = (u + v) mod p ; clobbers ta, tb
        if small:
            ins.append(Ins(f"v_add_u32 {ta}, {u}, {v}", [u, v], [ta]))
            ins.append(Ins(f"v_subrev_u32 {tb}, {P}, {ta}", [ta], [tb]))
            ins.append(Ins(f"v_min_u32 {out}, {ta}, {tb}", [ta, tb], [out]))
        else:
            ins.append(Ins(f"v_add_co_u32 {ta}, {sa}, {u}, {v}", [u, v], [ta, sa]))
            ins.append(Ins(f"v_subrev_co_u32 {tb}, {sb}, {P}, {ta}", [ta], [tb, sb]))
            ins.append(Ins(f"s_orn2_b64 {sa}, {sa}, {sb}", [sa, sb], [sa], salu=True))  # carry | !borrow
            ins.append(Ins(f"v_cndmask_b32 {out}, {ta}, {tb}, {sa}", [ta, tb, sa], [out]))

    def sub(u, v, out, tmp):  # out = (u - v) mod p ; clobbers tmp
        if small:
            ins.append(Ins(f"v_sub_u32 {out}, {u}, {v}", [u, v], [out]))
            ins.append(Ins(f"v_add_u32 {tmp}, {P}, {out}", [out], [tmp]))
            ins.append(Ins(f"v_min_u32 {out}, {out}, {tmp}", [out, tmp], [out]))
        else:
            ins.append(Ins(f"v_sub_co_u32 {out}, {sb}, {u}, {v}", [u, v], [out, sb]))
            ins.append(Ins(f"v_add_u32 {tmp}, {P}, {out}", [out], [tmp]))
            ins.append(Ins(f"v_cndmask_b32 {out}, {out}, {tmp}, {sb}", [out, tmp, sb], [out]))

    def mul(m, out, t1, t2):  # out = m * T * 2^-32 mod p, canonical ; clobbers t1, t2
        # Montgomery step as a second multiply-ADD: u = lo(m*T) * (-p^-1), then m*T + u*p has a zero low half and its
        # high half (plus the carry-out when p >= 2^31) is the result in [0, 2p): one v_mad_u64_u32 instead of
        # v_mul_hi_u32 + subtract + add.
        ins.append(Ins(f"v_mad_u64_u32 {LP}, vcc, {m}, {t}, 0", [m, t], [L[0], L[1], "vcc"]))
        ins.append(Ins(f"v_mul_lo_u32 {t1}, {L[0]}, {NPI}", [L[0]], [t1]))
        if small:  # 2p < 2^32: no carry
            ins.append(Ins(f"v_mad_u64_u32 {LP}, vcc, {t1}, {P}, {LP}", [t1, L[0], L[1]], [L[0], L[1], "vcc"]))
            ins.append(Ins(f"v_subrev_u32 {t2}, {P}, {L[1]}", [L[1]], [t2]))
            ins.append(Ins(f"v_min_u32 {out}, {L[1]}, {t2}", [L[1], t2], [out]))
        else:
            ins.append(Ins(f"v_mad_u64_u32 {LP}, {sa}, {t1}, {P}, {LP}", [t1, L[0], L[1]], [L[0], L[1], sa]))
            ins.append(Ins(f"v_subrev_co_u32 {t2}, {sb}, {P}, {L[1]}", [L[1]], [t2, sb]))
            ins.append(Ins(f"s_orn2_b64 {sa}, {sa}, {sb}", [sa, sb], [sa], salu=True))  # carry | !borrow: r >= p
            ins.append(Ins(f"v_cndmask_b32 {out}, {L[1]}, {t2}, {sa}", [L[1], t2, sa], [out]))

    # ---- lazy forms: every value in [0, 2p), 4p < 2^32 ----
    def lazy_add(u, v, out):  # out = u + v reduced to [0, 2p) ; clobbers ta, tb
        ins.append(Ins(f"v_add_u32 {ta}, {u}, {v}", [u, v], [ta]))            # < 4p
        ins.append(Ins(f"v_subrev_u32 {tb}, {P2}, {ta}", [ta], [tb]))         # wraps (huge) when ta < 2p
        ins.append(Ins(f"v_min_u32 {out}, {ta}, {tb}", [ta, tb], [out]))

    def lazy_mul(m, t1):  # L[1] = m * T * 2^-32 in [0, 2p) for any m < 2^32 (m*T + u*p < 2^33 * p < 2^63) ; clobbers t1
        ins.append(Ins(f"v_mad_u64_u32 {LP}, vcc, {m}, {t}, 0", [m, t], [L[0], L[1], "vcc"]))
        ins.append(Ins(f"v_mul_lo_u32 {t1}, {L[0]}, {NPI}", [L[0]], [t1]))
        ins.append(Ins(f"v_mad_u64_u32 {LP}, vcc, {t1}, {P}, {LP}", [t1, L[0], L[1]], [L[0], L[1], "vcc"]))

    if kind == "mul32":  # x' = x * T, canonical whatever the mode (the N^-1 sweep of the scaled inverse: T = N^-1, wave-uniform)
        if mode == "lazy":
            lazy_mul(x, ta)                                                     # [0, 2p)
            ins.append(Ins(f"v_subrev_u32 {tb}, {P}, {L[1]}", [L[1]], [tb]))    # wraps (huge) when < p
            ins.append(Ins(f"v_min_u32 {x}, {L[1]}, {tb}", [L[1], tb], [x]))
        else:
            mul(x, x, ta, tb)
        return ins
    if mode == "lazy":
        if kind == "fwd32":    # x' = x + y ; y' = (x - y + 2p) * T
            ins.append(Ins(f"v_sub_u32 {tc}, {x}, {y}", [x, y], [tc]))
            ins.append(Ins(f"v_add_u32 {tc}, {P2}, {tc}", [tc], [tc]))          # in (0, 4p)
            lazy_add(x, y, x)
            lazy_mul(tc, ta)
            ins.append(Ins(f"v_mov_b32 {y}, {L[1]}", [L[1]], [y]))
        else:                  # w = y * T ; x' = x + w ; y' = x - w   (w stays in the product's high half)
            lazy_mul(y, ta)
            ins.append(Ins(f"v_sub_u32 {y}, {x}, {L[1]}", [x, L[1]], [y]))      # wraps (huge) when x < w
            ins.append(Ins(f"v_add_u32 {tb}, {P2}, {y}", [y], [tb]))            # then this one is x - w + 2p
            ins.append(Ins(f"v_min_u32 {y}, {y}, {tb}", [y, tb], [y]))
            lazy_add(x, L[1], x)
        return ins
    if kind == "fwd32":    # x' = x + y ; y' = (x - y) * T
        sub(x, y, tc, ta)
        add(x, y, x)
        mul(tc, y, ta, tb)
    elif kind == "inv32":  # w = y * T ; x' = x + w ; y' = x - w
        mul(y, tc, ta, tb)
        sub(x, tc, y, ta)
        add(x, tc, x)
    return ins



# ---- one-lane interpreter of the generated streams (tests/test_gl_asm_sim.py checks them against big integers) ----------
def simulate(lines, env):
    """Execute the instruction texts on a dict of 32-bit registers / 1-bit flag pairs.  Operand spellings: %[name], vN,
    v[a:b], s[a:b], vcc, integer literals.  Returns env."""
    import re
    M32 = 0xFFFFFFFF

    def rd(op):
        op = op.strip()
        if re.fullmatch(r"-?\d+|0x[0-9a-fA-F]+", op):
            return int(op, 0) & M32
        return env[op]

    def rd64(op):
        op = op.strip()
        if op == "0":
            return 0
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", op)
        if m:
            return env["v" + m.group(1)] | (env["v" + m.group(2)] << 32)
        return env[op]  # %[pp]: a 64-bit scalar

    def wr64(op, v):
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", op.strip())
        env["v" + m.group(1)] = v & M32
        env["v" + m.group(2)] = (v >> 32) & M32

    for line in lines:
        if line.startswith("s_nop"):
            continue
        op, rest = line.split(None, 1)
        a = [x.strip() for x in rest.split(",")]
        # re-join v[a:b] / s[a:b] operands (they contain no comma, nothing to do) -- operands are already whole
        if op == "v_mov_b32":
            env[a[0]] = rd(a[1])
        elif op in ("v_add_co_u32", "v_addc_co_u32"):
            cin = env[a[4]] if op == "v_addc_co_u32" else 0
            t = rd(a[2]) + rd(a[3]) + cin
            env[a[0]], env[a[1]] = t & M32, t >> 32
        elif op in ("v_sub_co_u32", "v_subb_co_u32"):
            cin = env[a[4]] if op == "v_subb_co_u32" else 0
            t = rd(a[2]) - rd(a[3]) - cin
            env[a[0]], env[a[1]] = t & M32, 1 if t < 0 else 0
        elif op in ("v_subrev_co_u32", "v_subbrev_co_u32"):
            cin = env[a[4]] if op == "v_subbrev_co_u32" else 0
            t = rd(a[3]) - rd(a[2]) - cin
            env[a[0]], env[a[1]] = t & M32, 1 if t < 0 else 0
        elif op == "v_mad_u64_u32":
            t = rd(a[2]) * rd(a[3]) + rd64(a[4])
            wr64(a[0], t & ((1 << 64) - 1))
            env[a[1]] = t >> 64
        elif op == "v_cndmask_b32":
            env[a[0]] = rd(a[2]) if env[a[3]] else rd(a[1])
        elif op == "v_cmp_le_u64":
            env[a[0]] = 1 if rd64(a[1]) <= rd64(a[2]) else 0
        elif op == "s_or_b64":
            env[a[0]] = env[a[1]] | env[a[2]]
        elif op == "s_andn2_b64":
            env[a[0]] = env[a[1]] & (1 - env[a[2]])
        elif op == "s_orn2_b64":
            env[a[0]] = env[a[1]] | (1 - env[a[2]])
        elif op == "v_mul_lo_u32":
            env[a[0]] = (rd(a[1]) * rd(a[2])) & M32
        elif op == "v_mul_hi_u32":
            env[a[0]] = (rd(a[1]) * rd(a[2])) >> 32
        elif op == "v_add3_u32":
            env[a[0]] = (rd(a[1]) + rd(a[2]) + rd(a[3])) & M32
        elif op == "v_add_u32":
            env[a[0]] = (rd(a[1]) + rd(a[2])) & M32
        elif op == "v_sub_u32":
            env[a[0]] = (rd(a[1]) - rd(a[2])) & M32
        elif op == "v_subrev_u32":
            env[a[0]] = (rd(a[2]) - rd(a[1])) & M32
        elif op == "v_min_u32":
            env[a[0]] = min(rd(a[1]), rd(a[2]))
        else:
            raise ValueError("simulate: unknown instruction " + line)
    return env


def stream(kind, nb=2, vbase=104, mode=None):
    """The scheduled instruction lines of one statement (what emit()/emit32() put into gl_asm.h)."""
    if kind in ("fwd32", "inv32", "mul32"):
        return schedule([butterfly32(kind, b, mode, vbase if vbase != 104 else M32_VBASE) for b in range(nb)])
    if kind in ("fwd64", "inv64", "mul64", "invs64"):
        return schedule([butterfly64(kind, b, vbase) for b in range(nb)])
    return schedule([butterfly(kind, b, vbase) for b in range(nb)])

def emit32(kind, nb, mode, vbase=M32_VBASE):
    lists = [butterfly32(kind, b, mode, vbase) for b in range(nb)]
    lines = schedule(lists)
    nops = sum(1 for l in lines if l.startswith("s_nop"))
    name = f"m32_{kind[:3]}{nb}_{mode}"
    what = {"lazy": "p < 2^30, values in [0, 2p)", "small": "p < 2^31", "any": "any odd p < 2^32"}[mode]
    if kind == "mul32":
        args = ", ".join(f"uint32_t &x{b}" for b in range(nb)) + ", uint32_t t"
    else:
        args = ", ".join(f"uint32_t &x{b}, uint32_t &y{b}, uint32_t t{b}" for b in range(nb))
    extra = ", uint32_t p2" if mode == "lazy" and kind != "mul32" else ""
    src = [f"// {kind} x{nb} ({what}): {len(lines)} instructions, {nops} s_nop",
           f"__device__ __forceinline__ void {name}({args}, uint32_t p, uint32_t pinv{extra}) {{"]
    for b in range(nb):
        src.append(f"    uint32_t a_{b}, b_{b}, c_{b};" if kind != "mul32" else f"    uint32_t a_{b}, b_{b};")
    src.append("    asm volatile(")
    for l in lines:
        src.append(f'        "{l}\\n\\t"')
    outs, ins_ = [], []
    for b in range(nb):
        if kind == "mul32":
            outs += [f'[x_{b}] "+v"(x{b})'] + [f'[{r}{b}] "=&v"({r}{b})' for r in ("a_", "b_")]
            ins_ += [f'[t_{b}] "s"(t)']  # the one wave-uniform multiplier
        else:
            outs += [f'[x_{b}] "+v"(x{b})', f'[y_{b}] "+v"(y{b})']
            outs += [f'[{r}{b}] "=&v"({r}{b})' for r in ("a_", "b_", "c_")]
            ins_ += [f'[t_{b}] "v"(t{b})']
    ins_ += ['[p] "s"(p)', '[npinv] "s"(0u - pinv)']
    if mode == "lazy" and kind != "mul32":
        ins_ += ['[p2] "s"(p2)']
    clob = ['"vcc"', '"scc"'] + [f'"v{r}"' for r in range(vbase, vbase + 2 * nb)]
    if mode == "any":
        clob += [f'"s{r}"' for r in range(84, 84 + 4 * nb)]
    src.append("        : " + ", ".join(outs))
    src.append("        : " + ", ".join(ins_))
    src.append("        : " + ", ".join(clob) + ");")
    src.append("}")
    return "\n".join(src), len(lines), nops


def schedule(lists):
    """Merge instruction lists (each in program order) keeping dependencies; SGPR RAW pairs
    SGPR_DIST slots apart.  Returns lines (s_nop inserted only if unavoidable)."""
    nodes = []
    for li, lst in enumerate(lists):
        for k, ins in enumerate(lst):
            nodes.append((li, k, ins))
    n = len(nodes)
    preds = [[] for _ in range(n)]  # (pred index, min distance)
    for j in range(n):
        lj, kj, ij = nodes[j]
        for i in range(n):
            li, ki, ii = nodes[i]
            if li != lj or ki >= kj:
                continue
            dist = 0
            raw = ii.writes & ij.reads
            if raw:
                # the two-wait-state hazard is VALU writes SGPR/VCC -> VALU reads it; the scalar unit interlocks
                # (hipcc itself emits v_cmp + s_and_saveexec and s_mov + v_add back to back)
                hazard = any(is_sgpr(r) for r in raw) and not ii.salu and not ij.salu
                dist = max(dist, SGPR_DIST if hazard else 1)
            if (ii.reads & ij.writes) or (ii.writes & ij.writes):
                dist = max(dist, 1)
            if dist:
                preds[j].append((i, dist))
    # priority = longest path to the end
    succs = [[] for _ in range(n)]
    for j in range(n):
        for i, d in preds[j]:
            succs[i].append((j, d))
    prio = [0] * n
    for i in reversed(range(n)):
        prio[i] = max([d + prio[j] for j, d in succs[i]], default=0)
    slot = [None] * n
    out, t, done = [], 0, 0
    while done < n:
        best = None
        for j in range(n):
            if slot[j] is not None:
                continue
            if all(slot[i] is not None and slot[i] + d <= t for i, d in preds[j]):
                if best is None or prio[j] > prio[best]:
                    best = j
        if best is None:
            out.append("s_nop 0")
        else:
            slot[best] = t
            out.append(nodes[best][2].text)
            done += 1
        t += 1
    return out


def emit(kind, nb, tw_constraint, vbase=104, suffix="", sbase=80):
    lists = [butterfly(kind, b, vbase, sbase) for b in range(nb)]
    lines = schedule(lists)
    nops = sum(1 for l in lines if l.startswith("s_nop"))
    name = f"gl_{kind}{nb}_{'s' if tw_constraint == 's' else 'v'}{suffix}"
    args = []
    for b in range(nb):
        if kind == "mul":
            args += [f"uint64_t &x{b}", f"uint64_t t{b}"]
        else:
            args += [f"uint64_t &x{b}", f"uint64_t &y{b}", f"uint64_t t{b}"]
    if kind == "invs":
        args += ["uint64_t c"]
    src = [f"// {kind} x{nb}: {len(lines)} instructions, {nops} s_nop",
           f"__device__ __forceinline__ void {name}({', '.join(args)}) {{"]
    if kind == "invs":
        src.append("    const uint32_t c0 = (uint32_t) c, c1 = (uint32_t) (c >> 32);")
    for b in range(nb):
        src.append(f"    uint32_t x0_{b} = (uint32_t) x{b}, x1_{b} = (uint32_t) (x{b} >> 32);")
        if kind != "mul":
            src.append(f"    uint32_t y0_{b} = (uint32_t) y{b}, y1_{b} = (uint32_t) (y{b} >> 32);")
        src.append(f"    const uint32_t t0_{b} = (uint32_t) t{b}, t1_{b} = (uint32_t) (t{b} >> 32);")
        if kind != "mul":
            src.append(f"    uint32_t d0_{b}, d1_{b};")
    zero_regs = [vbase + 12 * b + 5 for b in range(nb)]  # A[1] of every butterfly slot
    for b, r in enumerate(zero_regs):
        # defined by an (identical, side-effect-free) asm so that the compiler merges the definitions of all
        # butterflies into one and cannot re-materialise the constant in front of every statement
        src.append(f"    register uint32_t zero_{b} asm(\"v{r}\");  // high half of the 64-bit addend {{x, 0}} of v_mad_u64_u32")
        src.append(f"    asm(\"v_mov_b32 %0, 0\" : \"=v\"(zero_{b}));")
    src.append("    asm volatile(")
    for l in lines:
        src.append(f'        "{l}\\n\\t"')
    outs, ins_ = [], []
    for b in range(nb):
        outs += [f'[x0_{b}] "+v"(x0_{b})', f'[x1_{b}] "+v"(x1_{b})']
        if kind != "mul":
            outs += [f'[y0_{b}] "+v"(y0_{b})', f'[y1_{b}] "+v"(y1_{b})']
            outs += [f'[{r}{b}] "=&v"({r}{b})' for r in ("d0_", "d1_")]
        ins_ += [f'[t0_{b}] "{tw_constraint}"(t0_{b})', f'[t1_{b}] "{tw_constraint}"(t1_{b})']
    if kind in ("fwd", "invs"):
        ins_ += ['[pp] "s"(0xFFFFFFFF00000001ull)']  # p, for the 64-bit compare of the modular add
    if kind == "invs":
        ins_ += ['[c0] "s"(c0)', '[c1] "s"(c1)']
    ins_ += [f'[zero_{b}] "v"(zero_{b})' for b in range(nb)]
    clob = ['"vcc"', '"scc"'] + [f'"v{r}"' for r in range(vbase, vbase + 12 * nb) if r not in zero_regs] + [f'"s{r}"' for r in range(sbase, sbase + 10 * nb)]
    src.append("        : " + ", ".join(outs))
    src.append("        : " + ", ".join(ins_))
    src.append("        : " + ", ".join(clob) + ");")
    for b in range(nb):
        src.append(f"    x{b} = ((uint64_t) x1_{b} << 32) | x0_{b};")
        if kind != "mul":
            src.append(f"    y{b} = ((uint64_t) y1_{b} << 32) | y0_{b};")
    src.append("}")
    return "\n".join(src), len(lines), nops


def emit64(kind, nb, tw_constraint, vbase=104, suffix="", vp=(102, 103)):
    """One M64 statement.  vp: the two VGPRs pinned to p's halves (outside the 24 scratch registers of the two slots)."""
    lists = [butterfly64(kind, b, vbase) for b in range(nb)]
    lines = schedule(lists)
    nops = sum(1 for l in lines if l.startswith("s_nop"))
    name = f"m64_{kind[:-2]}{nb}_{'s' if tw_constraint == 's' else 'v'}{suffix}"
    args = []
    for b in range(nb):
        if kind == "mul64":
            args += [f"uint64_t &x{b}", f"uint64_t t{b}"]
        else:
            args += [f"uint64_t &x{b}", f"uint64_t &y{b}", f"uint64_t t{b}"]
    if kind == "invs64":
        args += ["uint64_t c"]
    args += ["uint64_t p", "uint64_t pinv"]
    valu = sum(1 for l in lines if l.startswith("v_"))
    src = [f"// {kind} x{nb} (any odd p < 2^64): {len(lines)} instructions ({valu} VALU), {nops} s_nop",
           f"__device__ __forceinline__ void {name}({', '.join(args)}) {{"]
    for b in range(nb):
        src.append(f"    uint32_t x0_{b} = (uint32_t) x{b}, x1_{b} = (uint32_t) (x{b} >> 32);")
        if kind != "mul64":
            src.append(f"    uint32_t y0_{b} = (uint32_t) y{b}, y1_{b} = (uint32_t) (y{b} >> 32);")
        src.append(f"    const uint32_t t0_{b} = (uint32_t) t{b}, t1_{b} = (uint32_t) (t{b} >> 32);")
        if kind != "mul64":
            src.append(f"    uint32_t d0_{b}, d1_{b};")
    src.append("    const uint32_t p0 = (uint32_t) p, p1 = (uint32_t) (p >> 32), pi0 = (uint32_t) pinv, pi1 = (uint32_t) (pinv >> 32);")
    if kind == "invs64":
        src.append("    const uint32_t c0 = (uint32_t) c, c1 = (uint32_t) (c >> 32);")
    zero_regs = [vbase + 12 * b + 5 for b in range(nb)]
    for b, r in enumerate(zero_regs):
        src.append(f"    register uint32_t zero_{b} asm(\"v{r}\");  // high half of the 64-bit addend {{x, 0}} of v_mad_u64_u32")
        src.append(f"    asm(\"v_mov_b32 %0, 0\" : \"=v\"(zero_{b}));")
    # p in two pinned VGPRs, defined by side-effect-free asm: one pair of moves per kernel, not per statement
    src.append(f"    register uint32_t vp0 asm(\"v{vp[0]}\"), vp1 asm(\"v{vp[1]}\");")
    src.append("    asm(\"v_mov_b32 %0, %1\" : \"=v\"(vp0) : \"s\"(p0));")
    src.append("    asm(\"v_mov_b32 %0, %1\" : \"=v\"(vp1) : \"s\"(p1));")
    src.append("    asm volatile(")
    for l in lines:
        src.append(f'        "{l}\\n\\t"')
    outs, ins_ = [], []
    for b in range(nb):
        outs += [f'[x0_{b}] "+v"(x0_{b})', f'[x1_{b}] "+v"(x1_{b})']
        if kind != "mul64":
            outs += [f'[y0_{b}] "+v"(y0_{b})', f'[y1_{b}] "+v"(y1_{b})']
            outs += [f'[{r}{b}] "=&v"({r}{b})' for r in ("d0_", "d1_")]
        ins_ += [f'[t0_{b}] "{tw_constraint}"(t0_{b})', f'[t1_{b}] "{tw_constraint}"(t1_{b})']
    ins_ += ['[p0] "s"(p0)', '[p1] "s"(p1)', '[pi0] "s"(pi0)', '[pi1] "s"(pi1)', '[vp0] "v"(vp0)', '[vp1] "v"(vp1)']
    if kind == "invs64":
        ins_ += ['[c0] "s"(c0)', '[c1] "s"(c1)']
    ins_ += [f'[zero_{b}] "v"(zero_{b})' for b in range(nb)]
    clob = ['"vcc"', '"scc"'] + [f'"v{r}"' for r in range(vbase, vbase + 12 * nb) if r not in zero_regs] + [f'"s{r}"' for r in range(80, 80 + 10 * nb)]
    src.append("        : " + ", ".join(outs))
    src.append("        : " + ", ".join(ins_))
    src.append("        : " + ", ".join(clob) + ");")
    for b in range(nb):
        src.append(f"    x{b} = ((uint64_t) x1_{b} << 32) | x0_{b};")
        if kind != "mul64":
            src.append(f"    y{b} = ((uint64_t) y1_{b} << 32) | y0_{b};")
    src.append("}")
    return "\n".join(src), len(lines), nops


def main():
    out = ["// gl_asm.h -- GENERATED by tools/gen_gl_asm.py; do not edit.",
           "// Hand-scheduled gfx950 Goldilocks butterflies (two interleaved per statement); the",
           "// portable definition of the same arithmetic is FieldGL in field.h, and the GPU parity",
           "// tests hold the two bit-for-bit equal.  Replaces src/aie_core.cc:41-125.",
           "#pragma once", "#include <stdint.h>", "#if defined(__HIP_DEVICE_COMPILE__)", "namespace ntt {", ""]
    for kind in ("fwd", "inv", "mul"):
        for tw in ("v", "s"):
            txt, n, nops = emit(kind, 2, tw)
            out.append(txt)
            out.append("")
            print(f"{kind} x2 tw={tw}: {n} instructions, {nops} nops", file=sys.stderr)
        if kind == "mul":  # product of two data words inside the light kernels (fused polynomial product)
            txt, n, nops = emit(kind, 2, "v", vbase=72, suffix="_lo")
            out.append(txt)
            out.append("")
        if kind == "inv":  # the scaled last stage of the inverse transform (stage 0: its twiddles differ per thread -> "v" only)
            for vb, sfx in ((104, ""), (72, "_lo")):
                txt, n, nops = emit("invs", 2, "v", vbase=vb, suffix=sfx)
                out.append(txt)
                out.append("")
            print(f"invs x2: {n} instructions, {nops} nops", file=sys.stderr)
        if kind != "mul":
            for tw in ("v", "s"):  # for the radix-8 (light) kernels: scratch lives lower
                txt, n, nops = emit(kind, 2, tw, vbase=72, suffix="_lo")
                out.append(txt)
                out.append("")
            if os.environ.get("NTT_GEN_W") and kind == "fwd":  # EXPERIMENT (r05 8-wave bound): scratch at v[40:63] for a 64-VGPR kernel
                txt, n, nops = emit(kind, 2, "v", vbase=40, suffix="_w", sbase=int(os.environ.get("NTT_GEN_W_SBASE", "50")))
                out.append(txt)
                out.append("")
    for kind in ("fwd32", "inv32", "mul32"):
        for mode in ("lazy", "small", "any"):
            txt, n, nops = emit32(kind, 4, mode)
            out.append(txt)
            out.append("")
            print(f"{kind} x4 {mode}: {n} instructions, {nops} nops", file=sys.stderr)
    # general odd 64-bit modulus: heavy kernels (radix-16: scratch v[104:127], p in v[102:103]) and light ones (radix-8: v[72:95], v[96:97])
    for kind in ("fwd64", "inv64", "mul64"):
        for tw in ("v", "s"):
            for vb, sfx, vp in ((104, "", (102, 103)), (72, "_lo", (96, 97))):
                txt, n, nops = emit64(kind, 2, tw, vbase=vb, suffix=sfx, vp=vp)
                out.append(txt)
                out.append("")
        print(f"{kind} x2: {n} instructions, {nops} nops", file=sys.stderr)
    for vb, sfx, vp in ((104, "", (102, 103)), (72, "_lo", (96, 97))):  # scaled last inverse stage (stage-0 twiddles differ per thread: "v")
        txt, n, nops = emit64("invs64", 2, "v", vbase=vb, suffix=sfx, vp=vp)
        out.append(txt)
        out.append("")
    print(f"invs64 x2: {n} instructions, {nops} nops", file=sys.stderr)
    out += ["}  // namespace ntt", "#endif"]
    open(sys.argv[1] if len(sys.argv) > 1 else "ntt_aie_amd/csrc/gl_asm.h", "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
