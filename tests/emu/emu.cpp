// emu.cpp -- host index model of the pass kernels.
//
// TEST INFRASTRUCTURE.  Compiles the very same pass.h / field.h / plan.h the HIP
// kernels are built from with g++ and steps all 256 thread contexts of every
// workgroup phase by phase (the role __syncthreads() plays on the GPU), so the
// index rules, the LDS exchange pattern, the pass planner and the modular
// arithmetic can be checked against the oracle on a machine without a GPU.
// It is not a CPU fallback: nothing in the product loads this library.
#include <stdint.h>
#include <string.h>

#include <vector>

#define NTT_EMU_TRACK 1
#include <stdio.h>
#include <stdlib.h>

#include "../../ntt_aie_amd/csrc/pass.h"
#include "../../ntt_aie_amd/csrc/plan.h"

using namespace ntt;
using namespace ntt::host;

// Which template families this translation unit instantiates (bit set = compiled in; a call into an absent family returns
// EMU_ABSENT).  The ordinary build has them all; tests/test_emu_asan.py compiles one family per sanitizer executable so that
// the instrumented builds run in parallel and finish in about a minute instead of six.
//   0 Goldilocks forward   1 Goldilocks inverse   2 general 64-bit forward   3 general 64-bit inverse
//   4 4-byte forward       5 4-byte inverse       6 / 7 / 8 fused product middle: Goldilocks / general 64-bit / 4-byte
#ifndef EMU_PARTS
#define EMU_PARTS 0x1FF
#endif
#define EMU_HAS(bit) (((EMU_PARTS) >> (bit)) & 1)
enum { EMU_ABSENT = -100 };
// non-zero: the LDS hazard tracker stays off (emu_set_tracking; the sanitizer sweep checks memory safety, test_emu.py hazards)
static int g_no_track = 0;

namespace {

// LDS hazard tracker (pass.h: NTT_LDS_ACCESS).  epoch = number of WORKGROUP barriers so far.  A word may be read by a wave
// only if its last write is this wave's own or older than the last barrier; it may be written only if, in addition, every
// read of it since the last barrier was this wave's own.  Wave-local syncs do not advance the epoch: LDS operations of one
// wave execute in order, so same-wave accesses are always fine.  A violation aborts the process (the test then fails).
struct EmuLdsTrack : ntt::LdsTrack {
    struct St {
        int w_wave = -1, w_epoch = -1, r_wave = -1, r_epoch = -1;  // r_wave -2: several waves read it in r_epoch
    };
    std::vector<St> st;
    const char *base = nullptr;
    size_t word_bytes = 1;
    int epoch = 0;
    const char *what = "";
    void reset(const void *tile, size_t words, size_t wb) {  // a new workgroup
        base = (const char *) tile;
        word_bytes = wb;
        st.assign(words, St());
        epoch = 0;
    }
    void barrier() { ++epoch; }
    void access(const void *word, uint32_t tid, bool write) override {
        const int wave = (int) (tid >> 6);
        const size_t idx = (size_t) ((const char *) word - base) / word_bytes;
        if ((const char *) word < base || idx >= st.size()) return;  // not the tile (the product pass's twiddle tables)
        St &s = st[idx];
        const bool raw = s.w_epoch == epoch && s.w_wave != wave && s.w_wave != -1;
        const bool war = write && s.r_epoch == epoch && s.r_wave != wave && s.r_wave != -1;
        if (raw || war) {
            fprintf(stderr, "LDS hazard in %s: wave %d %s a word that wave %d %s since the last workgroup barrier (epoch %d)\n", what, wave,
                    write ? "writes" : "reads", raw ? s.w_wave : s.r_wave, raw ? "wrote" : "read", epoch);
            abort();
        }
        if (write) {
            s.w_wave = wave;
            s.w_epoch = epoch;
        } else if (s.r_epoch == epoch && s.r_wave != wave) {
            s.r_wave = -2;
        } else {
            s.r_wave = wave;
            s.r_epoch = epoch;
        }
    }
};

template <class Cfg>
struct EmuExec {
    static constexpr bool early_ok = true;
    std::vector<Ctx<Cfg>> ctx;
    std::vector<typename Cfg::W> tile;
    uint32_t bx, by;
    EmuExec() : ctx(Cfg::NT), tile(Cfg::DMA ? 2 * Cfg::TILE_WORDS : Cfg::LDS_WORDS) {}
    void init(const PassArgs<Cfg> &a) {
        for (int t = 0; t < Cfg::NT; t++) phase_init<Cfg>(ctx[t], a, (uint32_t) t, bx, by);
    }
    void init_indices(const PassArgs<Cfg> &a) {
        for (int t = 0; t < Cfg::NT; t++) phase_init<Cfg, false>(ctx[t], a, (uint32_t) t, bx, by);
    }
    // only_wave >= 0: step just that wave's 64 lanes (used to prove WAVE_LOCAL passes never read
    // another wave's LDS words: the four waves are then run one after the other, start to finish)
    int only_wave = -1;
    template <class Fn>
    void each(Fn &&f) {
        const int lo = only_wave < 0 ? 0 : 64 * only_wave, hi = only_wave < 0 ? Cfg::NT : lo + 64;
        for (int t = lo; t < hi; t++) f(ctx[t]);
    }
    EmuLdsTrack tr;
    void sync(std::false_type) { tr.barrier(); }
    void sync(std::true_type) {}
    uint32_t pg_base() const { return ctx[0].pg_base; }
    int ppw() const { return ctx[0].ppw; }
    bool iter_begin(int) { return true; }
    void iter_done(int) {}
    void pass_done(int) {}
    typename Cfg::W *lds() { return tile.data(); }
};

// host twin of GpuProductExec (kernels_gl_product.hip): all contexts of the workgroup stepped phase by phase
template <class CI, class CF>
struct EmuProductExec {
    using W = typename CI::W;
    std::vector<Ctx<CI>> ci;
    std::vector<Ctx<CF>> cf;
    std::vector<W> keep, pre, tile, tab_i, tab_f;
    uint32_t bx, by;
    int only_wave = -1;
    EmuProductExec()
        : ci(CI::NT), cf(CI::NT), keep((size_t) CI::NT * CI::E), pre((size_t) CI::NT * CI::E), tile(CI::LDS_WORDS),
          tab_i(tw_table_words<CI>()), tab_f(tw_table_words<CF>()) {}
    void init(const PassArgs<CI> &aa, const PassArgs<CF> &af) {
        for (int t = 0; t < CI::NT; t++) {
            phase_init<CI>(ci[t], aa, (uint32_t) t, bx, by);
            phase_init<CF>(cf[t], af, (uint32_t) t, bx, by);
        }
    }
    int lo() const { return only_wave < 0 ? 0 : 64 * only_wave; }
    int hi() const { return only_wave < 0 ? CI::NT : 64 * only_wave + 64; }
    template <class Fn>
    void eachI(Fn &&f) { for (int t = lo(); t < hi(); t++) f(ci[t]); }
    template <class Fn>
    void eachF(Fn &&f) { for (int t = lo(); t < hi(); t++) f(cf[t]); }
    template <class Fn>
    void eachIF(Fn &&f) { for (int t = lo(); t < hi(); t++) f(ci[t], cf[t], &keep[(size_t) t * CI::E], &pre[(size_t) t * CI::E]); }
    EmuLdsTrack tr;
    void sync(std::false_type) { tr.barrier(); }
    void sync(std::true_type) {}
    uint32_t pg_base() const { return ci[0].pg_base; }
    int ppw() const { return ci[0].ppw; }
    W *lds() { return tile.data(); }
    W *tabI() { return tab_i.data(); }
    W *tabF() { return tab_f.data(); }
};

struct Erased {
    const void *in;
    void *out;
    const void *tw;
    const void *tw_sc;  // Goldilocks scaled inverse: stage-0 twiddles * N^-1 (N/2 words), as ntt_api.hip prepares them
    uint32_t p, pinv, r2;
    uint64_t p64, pinv64, r2_64;  // FieldM64
    int n, s0;
    uint32_t batch;
    int layout, do_scale;
    uint64_t scale;
    uint32_t target_wgs;
    const void *in2;
    uint64_t pw_scale;
    int variant;  // PassDesc::variant of the pass being run (pass_kernel.inc: 1 = 4-byte CONTIG 10..12 stages on 512 threads x 8 words)
};

template <class F>
F make_field(const Erased &e);
template <>
FieldGL make_field<FieldGL>(const Erased &) {
    return FieldGL{};
}
template <>
FieldM32 make_field<FieldM32>(const Erased &e) {
    return FieldM32{e.p, e.pinv, e.r2};
}
template <>
FieldM64 make_field<FieldM64>(const Erased &e) {
    return FieldM64{e.p64, e.pinv64, e.r2_64};
}

template <class PC>
int run_product_mid(int n, uint32_t batch, uint32_t target_wgs, const void *a_in_, const void *b_in_, void *out_,
                    const void *tw_inv_, const void *tw_fwd_, uint64_t pw_scale, const Erased &fe) {
    using CI = typename PC::CI;
    using CF = typename PC::CF;
    using W = typename CI::W;
    constexpr int LOG_M = CI::LOG_M;
    const W *a_in = (const W *) a_in_, *b_in = (const W *) b_in_, *tw_inv = (const W *) tw_inv_, *tw_fwd = (const W *) tw_fwd_;
    W *out = (W *) out_;
    PassGeom g = pass_geometry(n, 0, LOG_M, 0, CI::LOG_U, true, batch, target_wgs);
    PassArgs<CI> aa;
    memset((void *) &aa, 0, sizeof(aa));
    aa.in = a_in;
    aa.tw = tw_inv;
    aa.field = make_field<typename CI::F>(fe);
    aa.n = n;
    aa.batch = batch;
    aa.ppw = g.ppw;
    aa.tp = g.tp;
    aa.log_ul = g.log_ul;
    aa.log_uh = g.log_uh;
    aa.log_up = g.log_up;
    aa.pg_stride = 1;
    PassArgs<CI> ab = aa;
    ab.in = b_in;
    PassArgs<CF> af;
    memset((void *) &af, 0, sizeof(af));
    af.out = out;
    af.tw = tw_fwd;
    af.field = make_field<typename CF::F>(fe);
    af.n = n;
    af.batch = batch;
    af.ppw = g.ppw;
    af.tp = g.tp;
    af.log_ul = g.log_ul;
    af.log_uh = g.log_uh;
    af.log_up = g.log_up;
    af.pg_stride = 1;
    af.pw_scale = (W) pw_scale;
    EmuProductExec<CI, CF> ex;
    for (uint32_t by = 0; by < g.grid_y; by++)
        for (uint32_t bx = 0; bx < g.grid_x; bx++) {
            ex.bx = bx;
            ex.by = by;
            memset(ex.tile.data(), 0xA5, ex.tile.size() * sizeof(W));
            ex.tr.reset(ex.tile.data(), ex.tile.size(), sizeof(W));
            ex.tr.what = "product pass";
            ntt::lds_track() = g_no_track ? nullptr : &ex.tr;
            run_product_pass<CI, CF>(ex, aa, ab, af);
            ntt::lds_track() = nullptr;
        }
    return 0;
}


template <class Cfg>
int run_cfg(const Erased &e) {
    using W = typename Cfg::W;
    PassArgs<Cfg> a;
    a.in = (const W *) e.in;
    a.out = (W *) e.out;
    a.tw = (const W *) e.tw;
    a.tw_sc = nullptr;
    constexpr bool CAN_FOLD = fold_scale<Cfg>();
    const bool sc = CAN_FOLD && e.do_scale;  // the launcher's rule (pass_kernel.inc: launch_cfg)
    if (sc) {
        if (!e.tw_sc) return -2;
        a.tw_sc = (const W *) e.tw_sc;
    }
    a.field = make_field<typename Cfg::F>(e);
    a.n = e.n;
    a.s0 = e.s0;
    a.batch = e.batch;
    a.layout = e.layout;
    a.do_scale = e.do_scale;
    a.scale = (W) e.scale;
    a.dbg = 0;
    a.pg_stride = 1;
    a.in2 = (const W *) e.in2;
    a.pw_scale = (W) e.pw_scale;
    a.skip_if = nullptr;
    PassGeom g = pass_geometry(e.n, e.s0, Cfg::LOG_M, Cfg::LOG_C, Cfg::LOG_U, Cfg::CONTIG, e.batch, e.target_wgs, Cfg::PPW_CAP);
    a.ppw = g.ppw;
    a.tp = g.tp;
    a.log_ul = g.log_ul;
    a.log_uh = g.log_uh;
    a.log_up = g.log_up;
    EmuExec<Cfg> ex;
    for (uint32_t by = 0; by < g.grid_y; by++)
        for (uint32_t bx = 0; bx < g.grid_x; bx++) {
            ex.bx = bx;
            ex.by = by;
            // poison the tile: a read of a word nobody wrote this launch shows up as garbage
            memset(ex.tile.data(), 0xA5, ex.tile.size() * sizeof(W));
            ex.tr.reset(ex.tile.data(), ex.tile.size(), sizeof(W));
            ex.tr.what = Cfg::CONTIG ? "CONTIG pass" : "column pass";
            ntt::lds_track() = g_no_track ? nullptr : &ex.tr;
            auto go = [&]() {
                if constexpr (CAN_FOLD) {
                    if (sc) return run_pass<Cfg, EmuExec<Cfg>, -1, true>(ex, a);
                }
                return run_pass<Cfg>(ex, a);
            };
            if constexpr (Cfg::WAVE_LOCAL) {
                for (int w = 0; w < Cfg::NT / 64; w++) {
                    ex.only_wave = w;
                    go();
                    memset(ex.tile.data(), 0x5A, ex.tile.size() * sizeof(W));  // nothing may survive
                }
            } else {
                go();
            }
            ntt::lds_track() = nullptr;
        }
    return 0;
}

template <class F, bool INV>
int dispatch(bool contig, int log_m, const Erased &e) {
#define CASE_CONTIG(M)                                                                                \
    case M:                                                                                           \
        return run_cfg<PassCfg<F, M, 0, true, INV, contig_preload_mask(M, sizeof(typename F::W))>>(e);
#define CASE_COL(M) \
    case M:         \
        return run_cfg<ColPassCfg<F, M, INV>>(e);
    if (contig) {
        if (log_m == 13) return run_cfg<PassCfg<F, 13, 0, true, INV, sizeof(typename F::W) == 4 ? 0xF : 0x8, 4, 9>>(e);  // pass_kernel.inc: ContigCfg13
        if (log_m == 14) {  // pass_kernel.inc: ContigCfg14 (4-byte words only)
            if constexpr (sizeof(typename F::W) == 4) return run_cfg<PassCfg<F, 14, 0, true, INV, 0xF, 4, 10>>(e);
            else return -1;
        }
        // pass_kernel.inc: the wide radix-8 variant of the single-pass sizes 2^10 .. 2^12 (same conditions as there)
        if (e.variant == 1 && e.in2 == nullptr) {
            if (log_m == 10) return run_cfg<PassCfg<F, 10, 0, true, INV, 0xF, 3, 9>>(e);
            if (log_m == 11) return run_cfg<PassCfg<F, 11, 0, true, INV, 0xF, 3, 9>>(e);
            if (log_m == 12) return run_cfg<PassCfg<F, 12, 0, true, INV, 0xF, 3, 9>>(e);
        }
        if (contig_log_e(log_m, sizeof(typename F::W), e.s0 + log_m == e.n) == 3) {
            if constexpr (!INV) {
                if (e.in2 != nullptr) {  // fused product: the non-DMA twins (pass_kernel.inc)
                    if (log_m == 7) return run_cfg<PassCfg<F, 7, 0, true, INV, 0xF, 3, 8, false>>(e);
                    if (log_m == 8) return run_cfg<PassCfg<F, 8, 0, true, INV, 0xF, 3, 8, false>>(e);
                    if (log_m == 9) return run_cfg<PassCfg<F, 9, 0, true, INV, 0xF, 3, 8, false>>(e);
                    if (log_m == 10) return run_cfg<PassCfg<F, 10, 0, true, INV, 0xF, 3, 9, false>>(e);
                    if (log_m == 11) return run_cfg<PassCfg<F, 11, 0, true, INV, 0xF, 3, 9, false>>(e);
                    return run_cfg<PassCfg<F, 12, 0, true, INV, 0xF, 3, 9, false>>(e);
                }
            }
            if (log_m == 7) return run_cfg<PassCfg<F, 7, 0, true, INV, 0xF, 3>>(e);
            if (log_m == 8) return run_cfg<PassCfg<F, 8, 0, true, INV, 0xF, 3>>(e);
            if (log_m == 9) return run_cfg<PassCfg<F, 9, 0, true, INV, 0xF, 3>>(e);
            if (log_m == 10) return run_cfg<PassCfg<F, 10, 0, true, INV, 0xF, 3, 9>>(e);
            if (log_m == 11) return run_cfg<PassCfg<F, 11, 0, true, INV, 0xF, 3, 9>>(e);
            return run_cfg<PassCfg<F, 12, 0, true, INV, 0xF, 3, 9>>(e);
        }
        switch (log_m) {
            CASE_CONTIG(1) CASE_CONTIG(2) CASE_CONTIG(3) CASE_CONTIG(4) CASE_CONTIG(5) CASE_CONTIG(6)
            CASE_CONTIG(7) CASE_CONTIG(8) CASE_CONTIG(9) CASE_CONTIG(10) CASE_CONTIG(11) CASE_CONTIG(12)
            default: return -1;
        }
    }
    switch (log_m) {
        CASE_COL(4) CASE_COL(5) CASE_COL(6) CASE_COL(7) CASE_COL(8) CASE_COL(9)
        default: return -1;
    }
}

// the pass of one family (see EMU_PARTS)
int dispatch_family(bool m64, int word_bytes, bool inverse, bool contig, int log_m, const Erased &e) {
    (void) contig; (void) log_m; (void) e;
    if (m64) {
#if EMU_HAS(2)
        if (!inverse) return dispatch<FieldM64, false>(contig, log_m, e);
#endif
#if EMU_HAS(3)
        if (inverse) return dispatch<FieldM64, true>(contig, log_m, e);
#endif
    } else if (word_bytes == 8) {
#if EMU_HAS(0)
        if (!inverse) return dispatch<FieldGL, false>(contig, log_m, e);
#endif
#if EMU_HAS(1)
        if (inverse) return dispatch<FieldGL, true>(contig, log_m, e);
#endif
    } else {
#if EMU_HAS(4)
        if (!inverse) return dispatch<FieldM32, false>(contig, log_m, e);
#endif
#if EMU_HAS(5)
        if (inverse) return dispatch<FieldM32, true>(contig, log_m, e);
#endif
    }
    return EMU_ABSENT;
}

}  // namespace

extern "C" {

void emu_set_tracking(int on) { g_no_track = !on; }

// Forward (inverse = 0) or exact inverse (inverse = 1, scaled by N^-1 when scale != 0)
// of `batch` polynomials, host buffers, table T in plain form (N words).
// passes_override: 0 = planner's split; otherwise a list "first,col,col,.." packed
// 4 bits each from the low nibble (used to exercise every tile shape); bits 60..63 = PassDesc::variant of the CONTIG pass.
int emu_transform(int word_bytes, int logn, uint64_t p, const void *T_plain, const void *in, void *out,
                  uint32_t batch, int inverse, int layout, int scale, uint32_t target_wgs,
                  uint64_t passes_override) {
    const size_t N = (size_t) 1 << logn;
    std::vector<uint64_t> T(N), Ti;
    for (size_t i = 0; i < N; i++)
        T[i] = word_bytes == 4 ? ((const uint32_t *) T_plain)[i] : ((const uint64_t *) T_plain)[i];
    if (inverse && !invert_table(T, p, Ti)) return -5;
    const std::vector<uint64_t> &src = inverse ? Ti : T;
    std::vector<uint32_t> t32;
    std::vector<uint64_t> t64;
    const void *tw;
    if (word_bytes == 4) {
        t32.resize(N);
        for (size_t i = 0; i < N; i++) t32[i] = (uint32_t) to_table_form(src[i], p, 4);
        tw = t32.data();
    } else {
        t64.resize(N);
        for (size_t i = 0; i < N; i++) t64[i] = to_table_form(src[i], p, 8);
        tw = t64.data();
    }
    std::vector<uint64_t> tsc;  // stage-0 twiddles of the scaled Goldilocks inverse, as ntt_plan_set_twiddles makes them
    if (inverse && word_bytes == 8) {
        const uint64_t ninv = powmod(p / 2 + 1, (uint64_t) logn, p);
        tsc.resize(N / 2);
        for (size_t i = 0; i < N / 2; i++) tsc[i] = to_table_form(mulmod(Ti[N / 2 + i], ninv, p), p, 8);
    }
    std::vector<PassDesc> passes;
    const int contig_variant = (int) (passes_override >> 60);
    passes_override &= (1ull << 60) - 1;
    if (passes_override == 0) {
        passes = plan_passes(logn, word_bytes);
    } else {
        int s0 = 0;
        bool first = true;
        for (uint64_t v = passes_override; v; v >>= 4) {
            int m = (int) (v & 15);
            passes.push_back({first, s0, m});
            s0 += m;
            first = false;
        }
        if (s0 != logn) return -1;
    }
    Erased e;
    memset(&e, 0, sizeof(e));
    e.p = (uint32_t) p;
    if (word_bytes == 4) {
        e.pinv = mont_pinv((uint32_t) p);
        e.r2 = mont_r2((uint32_t) p);
    }
    const bool m64 = word_bytes == 8 && p != GOLDILOCKS;  // general odd 64-bit modulus (ntt_api.hip: FK_M64)
    e.p64 = p;
    if (m64) {
        e.pinv64 = mont_pinv64(p);
        e.r2_64 = mont_r2_64(p);
    }
    e.n = logn;
    e.batch = batch;
    e.layout = layout;
    e.target_wgs = target_wgs;
    e.tw = tw;
    e.tw_sc = tsc.empty() ? nullptr : tsc.data();
    e.scale = to_table_form(powmod(p / 2 + 1, (uint64_t) logn, p), p, word_bytes);
    const void *cur = in;
    const size_t np = passes.size();
    for (size_t k = 0; k < np; k++) {
        const size_t i = inverse ? np - 1 - k : k;
        e.in = cur;
        e.out = out;
        e.s0 = passes[i].s0;
        e.variant = passes[i].contig ? (contig_variant ? contig_variant : passes[i].variant) : 0;
        e.do_scale = (inverse && scale && i == 0) ? 1 : 0;
        const int rc = dispatch_family(m64, word_bytes, inverse != 0, passes[i].contig, passes[i].log_m, e);
        if (rc) return rc;
        cur = out;
    }
    return 0;
}

// forward transform of (in * in2 * scale): the fused-product first pass (polymul's last leg)
int emu_forward_product(int word_bytes, int logn, uint64_t p, const void *T_plain, const void *in, const void *in2,
                        void *out, uint32_t batch, uint64_t scale, uint32_t target_wgs) {
    const size_t N = (size_t) 1 << logn;
    std::vector<uint32_t> t32(N);
    std::vector<uint64_t> t64(N);
    for (size_t i = 0; i < N; i++) {
        const uint64_t t = word_bytes == 4 ? ((const uint32_t *) T_plain)[i] : ((const uint64_t *) T_plain)[i];
        t32[i] = (uint32_t) to_table_form(t, p, 4 == word_bytes ? 4 : 8);
        t64[i] = to_table_form(t, p, 8);
    }
    Erased e;
    memset(&e, 0, sizeof(e));
    e.p = (uint32_t) p;
    if (word_bytes == 4) {
        e.pinv = mont_pinv((uint32_t) p);
        e.r2 = mont_r2((uint32_t) p);
    }
    e.n = logn;
    e.batch = batch;
    e.target_wgs = target_wgs;
    e.tw = word_bytes == 4 ? (const void *) t32.data() : (const void *) t64.data();
    const std::vector<PassDesc> passes = plan_passes(logn, word_bytes);
    const void *cur = in;
    for (size_t i = 0; i < passes.size(); i++) {
        e.in = cur;
        e.out = out;
        e.s0 = passes[i].s0;
        e.in2 = i == 0 ? in2 : nullptr;
        e.pw_scale = to_table_form(to_table_form(scale % p, p, word_bytes), p, word_bytes);
        const int rc = dispatch_family(false, word_bytes, false, passes[i].contig, passes[i].log_m, e);
        if (rc) return rc;
        cur = out;
    }
    return 0;
}

// The negacyclic product c = Fwd(InvU(a) . InvU(b) . N^-1) the way ntt_polymul_negacyclic runs it when the first pass
// has a product kernel: inverse column passes of both operands, the fused middle pass (pass.h: run_product_pass; the
// whole product for a single-pass size), forward column passes.  T_plain is the kind-2 table; a and b are overwritten
// (scratch), like on the device.
int emu_polymul_fused(int word_bytes, int logn, uint64_t p, const void *T_plain, void *a, void *b, void *out, uint32_t batch,
                      uint32_t target_wgs) {
    const size_t N = (size_t) 1 << logn;
    std::vector<uint64_t> T(N), Ti;
    for (size_t i = 0; i < N; i++) T[i] = word_bytes == 4 ? ((const uint32_t *) T_plain)[i] : ((const uint64_t *) T_plain)[i];
    if (!invert_table(T, p, Ti)) return -5;
    std::vector<uint64_t> tf64(N), ti64(N);
    std::vector<uint32_t> tf32(N), ti32(N);
    for (size_t i = 0; i < N; i++) {
        tf64[i] = to_table_form(T[i], p, word_bytes);
        ti64[i] = to_table_form(Ti[i], p, word_bytes);
        tf32[i] = (uint32_t) tf64[i];
        ti32[i] = (uint32_t) ti64[i];
    }
    const void *tf = word_bytes == 4 ? (const void *) tf32.data() : (const void *) tf64.data();
    const void *ti = word_bytes == 4 ? (const void *) ti32.data() : (const void *) ti64.data();
    const std::vector<PassDesc> passes = plan_passes(logn, word_bytes);
    const int m0 = passes[0].log_m;
    if (word_bytes == 8 ? (m0 < 7 || m0 > 12) : (m0 < 5 || m0 > 13)) return -1;  // unit sizes the product kernels exist for
    Erased e;
    memset(&e, 0, sizeof(e));
    e.p = (uint32_t) p;
    if (word_bytes == 4) {
        e.pinv = mont_pinv((uint32_t) p);
        e.r2 = mont_r2((uint32_t) p);
    }
    const bool m64 = word_bytes == 8 && p != GOLDILOCKS;
    e.p64 = p;
    if (m64) {
        e.pinv64 = mont_pinv64(p);
        e.r2_64 = mont_r2_64(p);
    }
    e.n = logn;
    e.batch = batch;
    e.target_wgs = target_wgs;
    for (size_t i = passes.size(); i-- > 1;)
        for (void *buf : {a, b}) {
            e.in = buf;
            e.out = buf;
            e.tw = ti;
            e.s0 = passes[i].s0;
            const int rc = dispatch_family(m64, word_bytes, true, passes[i].contig, passes[i].log_m, e);
            if (rc) return rc;
        }
    const uint64_t ninv = powmod(p / 2 + 1, (uint64_t) logn, p);
    const uint64_t pw = to_table_form(to_table_form(ninv, p, word_bytes), p, word_bytes);
    int rc = EMU_ABSENT;
    if (m64) {
#if EMU_HAS(7)
        switch (m0) {
#define PM(M) case M: rc = run_product_mid<ProductCfg<M, FieldM64>>(logn, batch, target_wgs, a, b, out, ti, tf, pw, e); break;
            PM(7) PM(8) PM(9) PM(10) PM(11) PM(12)
#undef PM
            default: return -1;
        }
#endif
    } else if (word_bytes == 8) {
#if EMU_HAS(6)
        switch (m0) {
#define PM(M) case M: rc = run_product_mid<ProductCfg<M>>(logn, batch, target_wgs, a, b, out, ti, tf, pw, e); break;
            PM(7) PM(8) PM(9) PM(10) PM(11) PM(12)
#undef PM
            default: return -1;
        }
#endif
    } else {
#if EMU_HAS(8)
        switch (m0) {
#define PM(M) case M: rc = run_product_mid<ProductCfgM32<M>>(logn, batch, target_wgs, a, b, out, ti, tf, pw, e); break;
            PM(5) PM(6) PM(7) PM(8) PM(9) PM(10) PM(11) PM(12) PM(13)
#undef PM
            default: return -1;
        }
#endif
    }
    if (rc) return rc;
    for (size_t i = 1; i < passes.size(); i++) {
        e.in = out;
        e.out = out;
        e.tw = tf;
        e.s0 = passes[i].s0;
        rc = dispatch_family(m64, word_bytes, false, passes[i].contig, passes[i].log_m, e);
        if (rc) return rc;
    }
    return 0;
}

// alternative `alt` of plan_alternatives(logn, word_bytes, p): writes up to 8 (contig, s0, log_m) triples, *min_batch;
// returns the number of passes, or -1 when there is no such alternative
int emu_plan_alt(int logn, int word_bytes, uint64_t p, int alt, int *out_triples, uint64_t *min_batch) {
    auto alts = plan_alternatives(logn, word_bytes, p);
    if (alt < 0 || alt >= (int) alts.size()) return -1;
    const auto &v = alts[(size_t) alt].passes;
    *min_batch = alts[(size_t) alt].min_batch;
    for (size_t i = 0; i < v.size() && i < 8; i++) {
        out_triples[3 * i] = v[i].contig;
        out_triples[3 * i + 1] = v[i].s0;
        out_triples[3 * i + 2] = v[i].log_m;
    }
    return (int) v.size();
}
// kernel variant (PassDesc::variant) of pass `pass` of alternative `alt`, or -1
int emu_plan_alt_variant(int logn, int word_bytes, uint64_t p, int alt, int pass) {
    auto alts = plan_alternatives(logn, word_bytes, p);
    if (alt < 0 || alt >= (int) alts.size() || pass < 0 || pass >= (int) alts[(size_t) alt].passes.size()) return -1;
    return alts[(size_t) alt].passes[(size_t) pass].variant;
}
int emu_select_alt(int logn, int word_bytes, uint64_t p, uint64_t batch) {
    return select_alternative(plan_alternatives(logn, word_bytes, p), batch);
}

// the planner's split, for tests: writes up to 8 (contig, s0, log_m) triples
int emu_plan(int logn, int word_bytes, int *out_triples) {
    auto v = plan_passes(logn, word_bytes);
    for (size_t i = 0; i < v.size() && i < 8; i++) {
        out_triples[3 * i] = v[i].contig;
        out_triples[3 * i + 1] = v[i].s0;
        out_triples[3 * i + 2] = v[i].log_m;
    }
    return (int) v.size();
}

// the launch geometry of one pass, for tests: out = {ppw, grid_x, grid_y, log_up, rows[0..3]}; returns the number of
// polynomial groups that the rule of phase_init() (the Ctx it fills for row by) covers other than exactly once
int emu_geometry(int n, int s0, int log_m, int log_c, int log_u, int contig, uint64_t batch, uint32_t target_wgs, int ppw_cap, uint32_t *out) {
    PassGeom g = pass_geometry(n, s0, log_m, log_c, log_u, contig != 0, batch, target_wgs, ppw_cap);
    out[0] = (uint32_t) g.ppw; out[1] = g.grid_x; out[2] = g.grid_y; out[3] = (uint32_t) g.log_up;
    for (int k = 0; k < 4; k++) out[4 + k] = g.tp.rows[k];
    using Cfg = PassCfg<FieldM32, 1, 0, true, false>;  // the row rule does not depend on the configuration
    PassArgs<Cfg> a;
    memset((void *) &a, 0, sizeof(a));
    a.ppw = g.ppw;
    a.tp = g.tp;
    a.n = n;
    const uint64_t groups = (batch + (1ull << g.log_up) - 1) >> g.log_up;
    std::vector<uint8_t> cov(groups, 0);
    for (uint32_t by = 0; by < g.grid_y; by++) {
        Ctx<Cfg> c;
        phase_init<Cfg, false>(c, a, 0, 0, by);
        if (c.ppw < 1) return -1;
        for (int it = 0; it < c.ppw; it++)
            if ((uint64_t) c.pg_base + it < groups && cov[(uint64_t) c.pg_base + it] < 255) cov[(uint64_t) c.pg_base + it]++;
    }
    int bad = 0;
    for (uint64_t i = 0; i < groups; i++) bad += cov[i] != 1;
    return bad;
}

// field arithmetic spot checks
uint64_t emu_gl_mul(uint64_t a, uint64_t b) { return FieldGL{}.mul_plain(a, b); }
uint64_t emu_gl_add(uint64_t a, uint64_t b) { return FieldGL{}.add(a, b); }
uint64_t emu_gl_sub(uint64_t a, uint64_t b) { return FieldGL{}.sub(a, b); }
uint64_t emu_m64_mul_plain(uint64_t a, uint64_t b, uint64_t p) {
    FieldM64 f{p, mont_pinv64(p), mont_r2_64(p)};
    return f.mul_plain(a, b);
}
uint64_t emu_m64_mul(uint64_t x, uint64_t tw, uint64_t p) { return FieldM64{p, mont_pinv64(p), mont_r2_64(p)}.mul(x, tw); }
uint64_t emu_m64_add(uint64_t a, uint64_t b, uint64_t p) { return FieldM64{p, 0, 0}.add(a, b); }
uint64_t emu_m64_sub(uint64_t a, uint64_t b, uint64_t p) { return FieldM64{p, 0, 0}.sub(a, b); }
uint32_t emu_m32_mul_plain(uint32_t a, uint32_t b, uint32_t p) {
    FieldM32 f{p, mont_pinv(p), mont_r2(p)};
    return f.mul_plain(a, b);
}
uint32_t emu_m32_add(uint32_t a, uint32_t b, uint32_t p) { return FieldM32{p, 0, 0}.add(a, b); }
uint32_t emu_m32_sub(uint32_t a, uint32_t b, uint32_t p) { return FieldM32{p, 0, 0}.sub(a, b); }

}  // extern "C"
