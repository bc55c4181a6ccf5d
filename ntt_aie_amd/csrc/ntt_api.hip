// ntt_api.hip -- C-ABI of libntt_hip.so (include/ntt_hip.h): plan, pass planner,
// twiddle preparation and launch sequencing.  Host side of what the reference
// does in src/test.cpp:62-190 (buffers, table, launch) minus XRT.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <memory>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/ntt_hip.h"
#include "guard.h"
#include "kernels.h"
#include "plan.h"

using namespace ntt::host;

namespace {

struct DeviceGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) err = hipSetDevice(dev);
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void) hipSetDevice(prev);
    }
};

// ROCTX ranges around every transform and every HBM pass when NTT_ROCTX=1 (rocprofv3 --marker-trace):
// the role of the reference's trace_event0() / trace_event1() brackets (src/aie_core.cc:129-131,
// src/aie2.py:182,312).  The ROCTX library is looked up at run time: no link-time profiler dependency.
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char *e = getenv("NTT_ROCTX");
        if (!e || atoi(e) == 0) return;
        // rocprofv3 listens to the rocprofiler-sdk ROCTX; libroctx64 is the older roctracer one (rocprof v1/v2)
        void *h = nullptr;
        for (const char *name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) return;
        push = (int (*)(const char *)) dlsym(h, "roctxRangePushA");
        pop = (int (*)()) dlsym(h, "roctxRangePop");
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
const Roctx &roctx() {
    static const Roctx r;
    return r;
}
struct RoctxRange {
    bool on;
    explicit RoctxRange(const char *name) : on(roctx().push != nullptr) {
        if (on) roctx().push(name);
    }
    RoctxRange(const char *what, int contig, int s0, int log_m) : on(roctx().push != nullptr) {
        if (!on) return;
        char buf[96];
        snprintf(buf, sizeof(buf), "%s %s stages %d-%d", what, contig ? "contig" : "column", s0, s0 + log_m - 1);
        roctx().push(buf);
    }
    ~RoctxRange() {
        if (on) roctx().pop();
    }
};

}  // namespace

struct ntt_plan {
    int logn;
    uint64_t p;
    int word_bytes;
    int device;
    int fk;  // arithmetic: FK_M32 (4-byte words), FK_GL (p = 2^64 - 2^32 + 1), FK_M64 (any other odd 8-byte modulus)
    // FieldM32 parameters
    uint32_t pinv, r2;
    // FieldM64 parameters
    uint64_t pinv64, r2_64;
    // device tables, table form
    void *d_tw_fwd;
    void *d_tw_inv;
    void *d_tw_inv_sc;  // 8-byte words (both fields): T^-1[N/2 + i] * N^-1, i < N/2 (stage-0 twiddles of the scaled inverse, pass.h: fold_scale)
    bool has_table, has_inv;
    uint64_t scale_tf;     // N^-1 in table form
    uint64_t ninv_plain;   // N^-1 plain
    uint32_t target_wgs;      // workgroups per launch the batch loop of a CONTIG pass is sized for
    uint32_t target_wgs_col;  // ... of a column pass (shorter loops win there)
    int dbg;            // experiment build only (NTT_DEBUG_FLAGS); always 0 in the product
    int force_variant;  // experiment build only (NTT_PASS_VARIANT=k): every CONTIG pass runs kernel variant k; -1 in the product
    int only_pass;      // experiment build only (NTT_ONLY_PASS=k): ntt_forward launches pass k alone (power / clock of one kernel); -1 in the product
    int fused;          // experiment build only (NTT_FUSED=1): N = 2^16 Goldilocks forward through the XCD-local fused launch
    void *d_fused_ctl;  // counters of the fused launch (plan-owned; null in the product)
    size_t fused_max_batch;
    unsigned long long *d_counter;  // one device word for ntt_count_noncanonical (no allocation per call)
    std::mutex counter_mu;          // ... which is the only entry point that writes plan-owned state after creation
    std::vector<PassDesc> passes;  // = alts[0].passes: the default decomposition (ntt_plan_info 3 / 32+ / 64+)
    std::vector<PlanAlt> alts;     // launch-time alternatives, ascending min_batch (plan.h: plan_alternatives)
    int forced_alt;                // ntt_plan_set_policy: -1 = by batch, k >= 0 = always alternative k
};

static_assert(NTT_E_NOMEM == NTT_E_NOMEM_GUARD && NTT_E_INTERNAL == NTT_E_INTERNAL_GUARD, "guard.h codes = include/ntt_hip.h codes");

namespace {

enum { FK_M32 = 0, FK_GL = 1, FK_M64 = 2 };

// frees what a (possibly half-built) plan owns; the device must be current
void free_plan(ntt_plan *pl) {
    if (!pl) return;
    if (pl->d_tw_fwd) (void) hipFree(pl->d_tw_fwd);
    if (pl->d_tw_inv) (void) hipFree(pl->d_tw_inv);
    if (pl->d_tw_inv_sc) (void) hipFree(pl->d_tw_inv_sc);
    if (pl->d_fused_ctl) (void) hipFree(pl->d_fused_ctl);
    if (pl->d_counter) (void) hipFree(pl->d_counter);
    delete pl;
}
// ntt_plan_create builds the plan under this holder: an exception (std::bad_alloc from the alternatives' vectors)
// unwinds through it, so the guard's error return leaks neither the struct nor device memory
struct PlanDeleter {
    void operator()(ntt_plan *pl) const {
        if (!pl) return;
        DeviceGuard g(pl->device);
        free_plan(pl);
    }
};
using PlanHolder = std::unique_ptr<ntt_plan, PlanDeleter>;

hipError_t launch_fwd(const ntt_plan *pl, const PassDesc &pd, const ntt::ErasedArgs &a, hipStream_t s) {
    return pl->fk == FK_GL ? ntt::launch_gl_fwd(pd.contig, pd.log_m, a, s)
         : pl->fk == FK_M64 ? ntt::launch_m64_fwd(pd.contig, pd.log_m, a, s)
                            : ntt::launch_m32_fwd(pd.contig, pd.log_m, a, s);
}
hipError_t launch_inv(const ntt_plan *pl, const PassDesc &pd, const ntt::ErasedArgs &a, hipStream_t s) {
    return pl->fk == FK_GL ? ntt::launch_gl_inv(pd.contig, pd.log_m, a, s)
         : pl->fk == FK_M64 ? ntt::launch_m64_inv(pd.contig, pd.log_m, a, s)
                            : ntt::launch_m32_inv(pd.contig, pd.log_m, a, s);
}

size_t table_bytes(const ntt_plan *pl) { return ((size_t) 1 << pl->logn) * pl->word_bytes; }
size_t sc_table_bytes(const ntt_plan *pl) { return pl->word_bytes == 8 ? table_bytes(pl) / 2 : 0; }

// the decomposition the launchers run for this batch
const std::vector<PassDesc> &passes_for(const ntt_plan *pl, size_t batch) {
    const int k = pl->forced_alt >= 0 ? pl->forced_alt : select_alternative(pl->alts, batch);
    return pl->alts[(size_t) k].passes;
}

#if defined(NTT_PHASE_STAMPS)
void *g_stamp_buf = nullptr;  // diagnostic build: ntt_stamps_set()
uint32_t g_stamp_records = 0;
#endif

ntt::ErasedArgs base_args(const ntt_plan *pl, const PassDesc &pd, const void *in, void *out, size_t batch) {
    ntt::ErasedArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in;
    a.out = out;
    a.p = (uint32_t) pl->p;
    a.pinv = pl->pinv;
    a.r2 = pl->r2;
    a.p64 = pl->p;
    a.pinv64 = pl->pinv64;
    a.r2_64 = pl->r2_64;
    a.n = pl->logn;
    a.s0 = pd.s0;
    a.batch = (uint32_t) batch;
    a.target_wgs = pd.contig ? pl->target_wgs : pl->target_wgs_col;
    a.dbg = pl->dbg;
    a.variant = pd.variant;
#if defined(NTT_EXPERIMENT)
    if (pl->force_variant >= 0 && pd.contig) a.variant = pl->force_variant;  // NTT_PASS_VARIANT=k: A/B of a kernel variant
#endif
#if defined(NTT_PHASE_STAMPS)
    // one region per pass kind, so that the passes of one transform do not overwrite each other's records: the CONTIG pass
    // takes the first half of the buffer, a column pass the second (tools/phase_stamps.py stamps two-pass transforms)
    a.stamp_records = g_stamp_records / 2;
    a.stamps = g_stamp_buf ? (char *) g_stamp_buf + (pd.contig ? 0 : (size_t) a.stamp_records * ntt::STAMP_RECORD * 8) : nullptr;
#endif
    return a;
}

// batch == 0 is a valid no-op whatever the pointers are (an empty torch tensor has a null data_ptr)
int check_io(const ntt_plan *pl, const void *a, const void *b, size_t batch) {
    if (!pl) return NTT_E_ARG;
    if (batch == 0) return NTT_OK;
    if (!a || !b) return NTT_E_ARG;
    if (((uintptr_t) a | (uintptr_t) b) & 15u) return NTT_E_ARG;  // 16-byte vector accesses
    if (batch > 0x7FFFFFFFull) return NTT_E_ARG;
    return NTT_OK;
}

// in2 != null: transform in[j] * in2[j] * pw_scale (plain) instead of in[j]; the product is folded into
// the load of the first pass
// `forced`: the decomposition to run (a product picks ONE for all of its transforms); null = passes_for(batch)
int run_forward(ntt_plan *pl, const void *d_in, void *d_out, size_t batch, int layout, hipStream_t s,
                const void *in2 = nullptr, uint64_t pw_scale_plain = 1, const std::vector<PassDesc> *forced = nullptr) {
    RoctxRange whole(in2 ? "ntt_forward(product)" : "ntt_forward");
    const void *src = d_in;
    const void *skip_if = nullptr;
#if defined(NTT_EXPERIMENT)
    if (pl->d_fused_ctl && !in2 && d_in != d_out && layout == NTT_LAYOUT_NATURAL && batch >= 64 && batch % 8 == 0 &&
        batch <= pl->fused_max_batch) {
        // one persistent XCD-local launch; the ordinary passes below then only run (device-side
        // decision, no host sync) if its check kernel did not certify the result
        hipError_t e = ntt::launch_fused_gl16(d_in, d_out, pl->d_tw_fwd, batch, pl->d_fused_ctl, s, pl->dbg);
        if (e != hipSuccess) return (int) e;
        skip_if = ntt::fused_gl16_ok_word(pl->d_fused_ctl);
        if (pl->dbg & (16 | 32 | 64)) return NTT_OK;  // timing experiments: fused launch alone
    }
#endif
    const std::vector<PassDesc> &passes = forced ? *forced : passes_for(pl, batch);
    for (const PassDesc &pd : passes) {
#if defined(NTT_EXPERIMENT)
        if (pl->only_pass >= 0 && (int) (&pd - &passes.front()) != pl->only_pass) continue;  // timing experiment: outputs meaningless
#endif
        RoctxRange pass("fwd pass", pd.contig, pd.s0, pd.log_m);
        ntt::ErasedArgs a = base_args(pl, pd, src, d_out, batch);
        a.skip_if = skip_if;
        if (&pd == &passes.front() && in2) {  // first pass only (d_out may alias d_in)
            a.in2 = in2;
            a.pw_scale = to_table_form(to_table_form(pw_scale_plain % pl->p, pl->p, pl->word_bytes), pl->p, pl->word_bytes);
        }
        a.tw = pl->d_tw_fwd;
        a.layout = layout;
        hipError_t e = launch_fwd(pl, pd, a, s);
        if (e != hipSuccess) return (int) e;
        src = d_out;
    }
    return NTT_OK;
}

int run_inverse(ntt_plan *pl, const void *d_in, void *d_out, size_t batch, int layout, int scale,
                hipStream_t s, const std::vector<PassDesc> *forced = nullptr) {
    RoctxRange whole("ntt_inverse");
    const void *src = d_in;
    const std::vector<PassDesc> &passes = forced ? *forced : passes_for(pl, batch);
    for (size_t i = passes.size(); i-- > 0;) {
        const PassDesc &pd = passes[i];
        RoctxRange pass("inv pass", pd.contig, pd.s0, pd.log_m);
        ntt::ErasedArgs a = base_args(pl, pd, src, d_out, batch);
        a.tw = pl->d_tw_inv;
        a.layout = layout;
        a.do_scale = (scale && i == 0) ? 1 : 0;
        a.scale = pl->scale_tf;
        // 8-byte words: N^-1 rides on the last executed stage (stage 0 of the CONTIG pass) instead of a sweep over the outputs
        a.tw_sc = a.do_scale ? pl->d_tw_inv_sc : nullptr;
        hipError_t e = launch_inv(pl, pd, a, s);
        if (e != hipSuccess) return (int) e;
        src = d_out;
    }
    return NTT_OK;
}

}  // namespace

extern "C" {

int ntt_version(void) { return 400; /* 0.4.0 */ }

#if defined(NTT_PHASE_STAMPS)
// Diagnostic side build only (tools/ab_build.sh stamps -DNTT_PHASE_STAMPS -> ab/libntt_stamps.so; tools/phase_stamps.py): where
// the pass kernels of this process write their phase stamps (pass.h: stamp()).  [records][ntt::STAMP_RECORD] 64-bit slots, one
// record per wave of a launch; null = stamps go to a dummy record.  Never declared in include/ntt_hip.h.
int ntt_stamps_set(void *d_buf, size_t records) NTT_GUARD {
    g_stamp_buf = d_buf;
    g_stamp_records = (uint32_t) (records > 0xFFFFFFFFull ? 0xFFFFFFFFull : records);
    return NTT_OK;
} NTT_GUARD_END
#endif

#if defined(NTT_EXPERIMENT)
// libntt_hip_exp.so only (never declared in include/ntt_hip.h, never exported by the product): the timing switches of
// PassArgs::dbg for ONE plan, set explicitly instead of through the process environment (bench.py's VALU-floor leg).
// The symbol doubles as the build's identity: ntt_aie_amd/_lib.py refuses a library that exports it.
int ntt_plan_set_debug(ntt_plan_t pl, int flags) NTT_GUARD {
    if (!pl) return NTT_E_ARG;
    pl->dbg = flags;
    return NTT_OK;
} NTT_GUARD_END
#endif

const char *ntt_error_string(int code) {
    switch (code) {
        case NTT_OK: return "ok";
        case NTT_E_ARG: return "invalid argument (null / misaligned pointer, size out of range)";
        case NTT_E_PRIME: return "unsupported modulus for this word size (must be odd, >= 3, and < 2^32 for 4-byte words)";
        case NTT_E_LOGN: return "logn out of range";
        case NTT_E_NOTABLE: return "twiddle table not set";
        case NTT_E_NOTINVERTIBLE: return "twiddle table has an entry that is not a unit mod p";
        case NTT_E_LAYOUT: return "AIE_BLOCK16 layout needs N >= 16";
        case NTT_E_RANGE: return "twiddle out of range [0, p)";
        case NTT_E_NODEVICE: return "no such HIP device";
        case NTT_E_NOMEM: return "out of host memory (a staging buffer of N words could not be allocated)";
        case NTT_E_INTERNAL: return "internal error (a C++ exception was caught at the C boundary)";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t) code);
    return "unknown error";
}

int ntt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ntt_plan_create(ntt_plan_t *out, int logn, uint64_t p, int word_bytes, int device) NTT_GUARD {
    if (!out) return NTT_E_ARG;
    *out = nullptr;
    if (word_bytes != 4 && word_bytes != 8) return NTT_E_ARG;
    if (logn < 1 || logn > NTT_MAX_LOGN) return NTT_E_LOGN;
    if ((p & 1) == 0 || p < 3) return NTT_E_PRIME;
    if (word_bytes == 4 && p > 0xFFFFFFFFull) return NTT_E_PRIME;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return NTT_E_NODEVICE;
    PlanHolder holder(new (std::nothrow) ntt_plan());
    ntt_plan *pl = holder.get();
    if (!pl) return NTT_E_NOMEM;
    pl->d_tw_fwd = pl->d_tw_inv = pl->d_tw_inv_sc = nullptr;
    pl->d_fused_ctl = nullptr;
    pl->d_counter = nullptr;
    pl->device = device;
    pl->logn = logn;
    pl->p = p;
    pl->word_bytes = word_bytes;
    pl->has_table = pl->has_inv = false;
    pl->pinv = pl->r2 = 0;
    pl->pinv64 = pl->r2_64 = 0;
    pl->fk = word_bytes == 4 ? FK_M32 : (p == GOLDILOCKS ? FK_GL : FK_M64);
    if (pl->fk == FK_M32) {
        pl->pinv = mont_pinv((uint32_t) p);
        pl->r2 = mont_r2((uint32_t) p);
    } else if (pl->fk == FK_M64) {
        pl->pinv64 = mont_pinv64(p);
        pl->r2_64 = mont_r2_64(p);
    }
    pl->ninv_plain = powmod(p / 2 + 1, (uint64_t) logn, p);  // (2^-1)^logn; 2^-1 = (p + 1) / 2 = p / 2 + 1 for odd p (no overflow at p near 2^64)
    pl->scale_tf = to_table_form(pl->ninv_plain, p, word_bytes);
    pl->target_wgs = 8192;  // workgroups per launch the batch loop is sized for (sweep: profiles/)
    // column passes: 16384 (their tile streams 8 polynomials per workgroup at N = 2^16, batch 4096, instead of 16): -4 %
    pl->target_wgs_col = 2 * pl->target_wgs;
    pl->dbg = 0;
    pl->only_pass = -1;
    pl->force_variant = -1;
    pl->fused = 0;
    pl->fused_max_batch = 0;
    pl->alts = plan_alternatives(logn, word_bytes, p);
    pl->passes = pl->alts[0].passes;
    pl->forced_alt = -1;
#if defined(NTT_EXPERIMENT)
    // Experiment knobs exist only in libntt_hip_exp.so (make exp; tools/): the product library reads NO environment
    // variable, so a stray NTT_DEBUG_FLAGS in a user's shell cannot redirect loads and stores.
    if (const char *e = getenv("NTT_TARGET_WGS")) {
        long v = atol(e);
        if (v > 0 && v < (1 << 24)) pl->target_wgs = (uint32_t) v, pl->target_wgs_col = 2 * pl->target_wgs;
    }
    if (const char *e = getenv("NTT_TARGET_WGS_COL")) {
        long v = atol(e);
        if (v > 0 && v < (1 << 24)) pl->target_wgs_col = (uint32_t) v;
    }
    if (const char *e = getenv("NTT_DEBUG_FLAGS")) pl->dbg = atoi(e);
    if (const char *e = getenv("NTT_ONLY_PASS")) pl->only_pass = atoi(e);
    if (const char *e = getenv("NTT_PASS_VARIANT")) pl->force_variant = atoi(e);
    if (const char *e = getenv("NTT_FUSED")) pl->fused = atoi(e);
    if (const char *e = getenv("NTT_PLAN_SPLIT")) {  // "8,6,6" = CONTIG 8 stages + two 6-stage column passes
        std::vector<PassDesc> v;
        int s0 = 0;
        bool ok = true;
        for (const char *c = e; *c && ok;) {
            char *end = nullptr;
            const long m = strtol(c, &end, 10);
            if (end == c) break;
            if (v.empty()) ok = m >= 1 && m <= (word_bytes == 4 ? 14 : 13);
            else ok = m >= MIN_COL_LOG_M && m <= MAX_COL_LOG_M_WIDE;
            v.push_back({v.empty(), s0, (int) m, 0});
            s0 += (int) m;
            c = *end == ',' ? end + 1 : end;
        }
        if (v.size() > 1 && v[0].log_m < ntt::col_log_c(word_bytes)) ok = false;  // column tiles are 2^log_c words wide
        if (ok && s0 == logn && !v.empty()) {  // anything else: keep the planner's alternatives
            pl->passes = v;
            pl->alts.assign(1, PlanAlt{v, 0});
        }
    }
#endif
    DeviceGuard g(device);
    if (g.err != hipSuccess) return (int) g.err;
    hipError_t e = hipMalloc(&pl->d_tw_fwd, table_bytes(pl));
    if (e == hipSuccess) e = hipMalloc(&pl->d_tw_inv, table_bytes(pl));
    if (e == hipSuccess && sc_table_bytes(pl)) e = hipMalloc(&pl->d_tw_inv_sc, sc_table_bytes(pl));
    if (e == hipSuccess) e = hipMalloc(&pl->d_counter, sizeof(*pl->d_counter));
#if defined(NTT_EXPERIMENT)
    if (e == hipSuccess && pl->fused && logn == 16 && word_bytes == 8) {
        pl->fused_max_batch = (size_t) 1 << 20;
        e = hipMalloc(&pl->d_fused_ctl, ntt::fused_gl16_ctl_bytes(pl->fused_max_batch));
    }
#endif
    if (e != hipSuccess) return (int) e;  // the holder frees what was allocated
    *out = holder.release();
    return NTT_OK;
} NTT_GUARD_END

int ntt_plan_destroy(ntt_plan_t pl) NTT_GUARD {
    if (!pl) return NTT_E_ARG;
    DeviceGuard g(pl->device);
    free_plan(pl);
    return NTT_OK;
} NTT_GUARD_END

int ntt_plan_set_twiddles(ntt_plan_t pl, const void *host_T) NTT_GUARD {
    if (!pl || !host_T) return NTT_E_ARG;
    const size_t N = (size_t) 1 << pl->logn;
    const uint64_t p = pl->p;
    std::vector<uint64_t> T(N), Ti(N, 0);
    for (size_t i = 0; i < N; i++)
        T[i] = pl->word_bytes == 4 ? ((const uint32_t *) host_T)[i] : ((const uint64_t *) host_T)[i];
    for (size_t i = 1; i < N; i++)
        if (T[i] >= p) return NTT_E_RANGE;
    T[0] %= p;  // T[0] is never read by the network (src/test.cpp:45 uses h+i >= 1)
    const bool inv_ok = invert_table(T, p, Ti);
    std::vector<unsigned char> buf_f(table_bytes(pl)), buf_i(table_bytes(pl));
    if (pl->word_bytes == 4) {
        for (size_t i = 0; i < N; i++) {
            ((uint32_t *) buf_f.data())[i] = (uint32_t) to_table_form(T[i], p, 4);
            ((uint32_t *) buf_i.data())[i] = (uint32_t) to_table_form(Ti[i], p, 4);
        }
    } else {
        for (size_t i = 0; i < N; i++) {
            ((uint64_t *) buf_f.data())[i] = to_table_form(T[i], p, 8);
            ((uint64_t *) buf_i.data())[i] = to_table_form(Ti[i], p, 8);
        }
    }
    DeviceGuard g(pl->device);
    if (g.err != hipSuccess) return (int) g.err;
    hipError_t e = hipMemcpy(pl->d_tw_fwd, buf_f.data(), buf_f.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(pl->d_tw_inv, buf_i.data(), buf_i.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess && pl->d_tw_inv_sc && inv_ok) {
        std::vector<uint64_t> sc(N / 2);
        for (size_t i = 0; i < N / 2; i++) sc[i] = to_table_form(mulmod(Ti[N / 2 + i], pl->ninv_plain, p), p, 8);
        e = hipMemcpy(pl->d_tw_inv_sc, sc.data(), sc.size() * sizeof(uint64_t), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) return (int) e;
    pl->has_table = true;
    pl->has_inv = inv_ok;
    return NTT_OK;
} NTT_GUARD_END

int ntt_make_table(ntt_plan_t pl, int kind, uint64_t g, void *host_T) NTT_GUARD {
    if (!pl || !host_T) return NTT_E_ARG;
    const uint64_t N = 1ull << pl->logn;
    std::vector<uint64_t> T;
    if (kind < 0 || kind > 2 || !make_table(kind, pl->logn, pl->p, g, T)) return NTT_E_ARG;
    for (uint64_t i = 0; i < N; i++) {
        if (pl->word_bytes == 4) ((uint32_t *) host_T)[i] = (uint32_t) T[i];
        else ((uint64_t *) host_T)[i] = T[i];
    }
    return NTT_OK;
} NTT_GUARD_END

int ntt_plan_generate_twiddles(ntt_plan_t pl, int kind, uint64_t g) NTT_GUARD {
    if (!pl || kind < 0 || kind > 2) return NTT_E_ARG;
    const uint64_t N = 1ull << pl->logn, p = pl->p;
    uint64_t base;
    if (kind == 2) {
        if ((p - 1) % (2 * N)) return NTT_E_ARG;
        base = invmod(powmod(g, (p - 1) / (2 * N), p), p);  // psi^-1
    } else {
        if (kind == 1 && (p - 1) % N) return NTT_E_ARG;
        base = powmod(g, (p - 1) / N, p);                           // w (integer division, src/test.cpp:28)
    }
    if (base == 0) return NTT_E_NOTINVERTIBLE;
    const uint64_t base_inv = invmod(base, p);
    if (base_inv == 0) return NTT_E_NOTINVERTIBLE;
    const int wb = pl->word_bytes;
    const uint64_t one_m = to_table_form(1 % p, p, wb);
    DeviceGuard g_(pl->device);
    if (g_.err != hipSuccess) return (int) g_.err;
    hipError_t e;
    if (pl->fk == FK_M64) {
        e = ntt::launch_gen_table_m64(pl->d_tw_fwd, pl->logn, kind, to_table_form(base, p, 8), one_m, p, pl->pinv64, pl->r2_64, nullptr);
        if (e == hipSuccess)
            e = ntt::launch_gen_table_m64(pl->d_tw_inv, pl->logn, kind, to_table_form(base_inv, p, 8), one_m, p, pl->pinv64, pl->r2_64, nullptr);
        if (e == hipSuccess)
            e = ntt::launch_scale_table_m64((const uint64_t *) pl->d_tw_inv + N / 2, pl->d_tw_inv_sc, N / 2, pl->scale_tf, p, pl->pinv64,
                                            pl->r2_64, nullptr);
    } else if (wb == 8) {
        e = ntt::launch_gen_table_gl(pl->d_tw_fwd, pl->logn, kind, to_table_form(base, p, 8), one_m, nullptr);
        if (e == hipSuccess)
            e = ntt::launch_gen_table_gl(pl->d_tw_inv, pl->logn, kind, to_table_form(base_inv, p, 8), one_m, nullptr);
        if (e == hipSuccess)  // stage-0 twiddles of the scaled inverse: T^-1[N/2 + i] * N^-1
            e = ntt::launch_scale_table_gl((const uint64_t *) pl->d_tw_inv + N / 2, pl->d_tw_inv_sc, N / 2, pl->scale_tf, nullptr);
    } else {
        e = ntt::launch_gen_table_m32(pl->d_tw_fwd, pl->logn, kind, (uint32_t) to_table_form(base, p, 4),
                                      (uint32_t) one_m, (uint32_t) p, pl->pinv, pl->r2, nullptr);
        if (e == hipSuccess)
            e = ntt::launch_gen_table_m32(pl->d_tw_inv, pl->logn, kind, (uint32_t) to_table_form(base_inv, p, 4),
                                          (uint32_t) one_m, (uint32_t) p, pl->pinv, pl->r2, nullptr);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) return (int) e;
    pl->has_table = true;
    pl->has_inv = true;  // every entry is a power of a unit
    return NTT_OK;
} NTT_GUARD_END

int ntt_plan_get_twiddles(ntt_plan_t pl, int inverse, void *host_T) NTT_GUARD {
    if (!pl || !host_T) return NTT_E_ARG;
    if (!pl->has_table) return NTT_E_NOTABLE;
    if (inverse && !pl->has_inv) return NTT_E_NOTINVERTIBLE;
    DeviceGuard g_(pl->device);
    if (g_.err != hipSuccess) return (int) g_.err;
    hipError_t e = hipMemcpy(host_T, inverse ? pl->d_tw_inv : pl->d_tw_fwd, table_bytes(pl), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return (int) e;
    // table (Montgomery) form -> plain residues: x * R^-1
    const size_t N = (size_t) 1 << pl->logn;
    const uint64_t p = pl->p;
    const uint64_t rinv = invmod(to_table_form(1 % p, p, pl->word_bytes), p);  // R is a power of two, p odd: always a unit
    for (size_t i = 0; i < N; i++) {
        if (pl->word_bytes == 4) ((uint32_t *) host_T)[i] = (uint32_t) mulmod(((uint32_t *) host_T)[i], rinv, p);
        else ((uint64_t *) host_T)[i] = mulmod(((uint64_t *) host_T)[i], rinv, p);
    }
    return NTT_OK;
} NTT_GUARD_END

int ntt_make_roots(ntt_plan_t pl, uint64_t g, void *host_T) NTT_GUARD { return ntt_make_table(pl, 0, g, host_T); } NTT_GUARD_END

int64_t ntt_plan_info(ntt_plan_t pl, int what) NTT_GUARD {
    if (!pl) return NTT_E_ARG;
    switch (what) {
        case 0: return pl->logn;
        case 1: return pl->word_bytes;
        case 2: return pl->device;
        case 3: return (int64_t) pl->passes.size();
        case 4: return pl->has_inv ? 1 : 0;
        case 5: return pl->d_fused_ctl ? 1 : 0;
        case 6: return (int64_t) pl->alts.size();
        case 7: return pl->forced_alt;
        case 8: {  // capacity for ntt_forward_profile whatever the batch
            size_t k = 0;
            for (const PlanAlt &a : pl->alts) k = a.passes.size() > k ? a.passes.size() : k;
            return (int64_t) k;
        }
        default: break;
    }
    if (what >= 256 && what < 256 + 16 * (int) pl->alts.size()) {
        const PlanAlt &alt = pl->alts[(size_t) (what - 256) / 16];
        const int k = (what - 256) % 16;
        if (k == 0) return (int64_t) alt.passes.size();
        if (k >= 1 && k <= 7) return k - 1 < (int) alt.passes.size() ? alt.passes[(size_t) k - 1].log_m : NTT_E_ARG;
        if (k >= 8 && k <= 14) return k - 8 < (int) alt.passes.size() ? alt.passes[(size_t) k - 8].s0 : NTT_E_ARG;
        return (int64_t) alt.min_batch;  // k == 15
    }
    if (what >= 512 && what < 512 + 16 * (int) pl->alts.size()) {  // kernel variant of pass k of alternative a (PassDesc::variant)
        const PlanAlt &alt = pl->alts[(size_t) (what - 512) / 16];
        const int k = (what - 512) % 16;
        return k < (int) alt.passes.size() ? alt.passes[(size_t) k].variant : NTT_E_ARG;
    }
    if (what >= 32 && what < 32 + (int) pl->passes.size()) return pl->passes[what - 32].log_m;
    if (what >= 64 && what < 64 + (int) pl->passes.size()) return pl->passes[what - 64].s0;
#if defined(NTT_EXPERIMENT)
    if (what >= 16 && what < 16 + 12 && pl->d_fused_ctl) {  // diagnostics: words of the last fused launch (blocking)
        uint32_t w[12];
        DeviceGuard g(pl->device);
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(w, pl->d_fused_ctl, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess)
            return NTT_E_ARG;
        return w[what - 16];  // 0-7 slots per XCC, 8 status, 9 b_done, 10 ok
    }
#endif
    return NTT_E_ARG;
} NTT_GUARD_END

int ntt_plan_select(ntt_plan_t pl, size_t batch) NTT_GUARD {
    if (!pl) return NTT_E_ARG;
    return pl->forced_alt >= 0 ? pl->forced_alt : select_alternative(pl->alts, batch);
} NTT_GUARD_END

int ntt_plan_set_policy(ntt_plan_t pl, int alternative) NTT_GUARD {
    if (!pl || alternative < -1 || alternative >= (int) pl->alts.size()) return NTT_E_ARG;
    pl->forced_alt = alternative;
    return NTT_OK;
} NTT_GUARD_END

int ntt_plan_clone(ntt_plan_t src, int device, ntt_plan_t *out) NTT_GUARD {
    if (!out) return NTT_E_ARG;
    *out = nullptr;
    if (!src) return NTT_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return NTT_E_NODEVICE;
    ntt_plan_t pl = nullptr;
    int rc = ntt_plan_create(&pl, src->logn, src->p, src->word_bytes, device);
    if (rc != NTT_OK) return rc;
    pl->alts = src->alts;  // an experiment split or a forced policy travels with the plan
    pl->passes = src->passes;
    pl->forced_alt = src->forced_alt;
    pl->target_wgs = src->target_wgs;
    pl->target_wgs_col = src->target_wgs_col;
    if (src->has_table) {
        // device-to-device, no host copy of the table: over xGMI when the devices differ (hipMemcpyPeer), which is what the
        // reference's on-chip table broadcast does below its host (src/aie2.py:96-104)
        auto copy = [&](void *dst, const void *from, size_t bytes) -> hipError_t {
            if (bytes == 0 || !dst || !from) return hipSuccess;
            return device == src->device ? hipMemcpy(dst, from, bytes, hipMemcpyDeviceToDevice)
                                         : hipMemcpyPeer(dst, device, from, src->device, bytes);
        };
        DeviceGuard g(device);
        hipError_t e = g.err;
        if (e == hipSuccess) e = copy(pl->d_tw_fwd, src->d_tw_fwd, table_bytes(src));
        if (e == hipSuccess) e = copy(pl->d_tw_inv, src->d_tw_inv, table_bytes(src));
        if (e == hipSuccess) e = copy(pl->d_tw_inv_sc, src->d_tw_inv_sc, sc_table_bytes(src));
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            (void) ntt_plan_destroy(pl);
            return (int) e;
        }
        pl->has_table = true;
        pl->has_inv = src->has_inv;
    }
    *out = pl;
    return NTT_OK;
} NTT_GUARD_END

int ntt_forward(ntt_plan_t pl, const void *d_in, void *d_out, size_t batch, int out_layout, void *stream) NTT_GUARD {
    int rc = check_io(pl, d_in, d_out, batch);
    if (rc) return rc;
    if (!pl->has_table) return NTT_E_NOTABLE;
    if (out_layout != NTT_LAYOUT_NATURAL && out_layout != NTT_LAYOUT_AIE_BLOCK16) return NTT_E_ARG;
    if (out_layout == NTT_LAYOUT_AIE_BLOCK16 && pl->logn < 4) return NTT_E_LAYOUT;
    if (batch == 0) return NTT_OK;
    DeviceGuard g(pl->device);
    if (g.err != hipSuccess) return (int) g.err;
    return run_forward(pl, d_in, d_out, batch, out_layout, (hipStream_t) stream);
} NTT_GUARD_END

int ntt_forward_profile(ntt_plan_t pl, const void *d_in, void *d_out, size_t batch, int out_layout,
                        void *stream, float *ms_per_pass, int max_passes, int *n_passes) NTT_GUARD {
    int rc = check_io(pl, d_in, d_out, batch);
    if (rc) return rc;
    if (!ms_per_pass || !n_passes) return NTT_E_ARG;
    const std::vector<PassDesc> &passes = passes_for(pl, batch);
    *n_passes = (int) passes.size();
    if (max_passes < (int) passes.size()) return NTT_E_ARG;
    if (!pl->has_table) return NTT_E_NOTABLE;
    if (batch == 0) return NTT_OK;
    DeviceGuard g(pl->device);
    if (g.err != hipSuccess) return (int) g.err;
    hipStream_t s = (hipStream_t) stream;
    const size_t np = passes.size();
    std::vector<hipEvent_t> ev;
    ev.reserve(np + 1);
    for (size_t i = 0; i <= np; i++) {
        hipEvent_t x;
        const hipError_t ce = hipEventCreate(&x);
        if (ce != hipSuccess) {  // destroy the ones already made
            for (auto &y : ev) (void) hipEventDestroy(y);
            return (int) ce;
        }
        ev.push_back(x);
    }
    const void *src = d_in;
    hipError_t e = hipEventRecord(ev[0], s);
    for (size_t i = 0; i < np && e == hipSuccess; i++) {
        const PassDesc &pd = passes[i];
        ntt::ErasedArgs a = base_args(pl, pd, src, d_out, batch);
        a.tw = pl->d_tw_fwd;
        a.layout = out_layout;
        e = launch_fwd(pl, pd, a, s);
        if (e == hipSuccess) e = hipEventRecord(ev[i + 1], s);
        src = d_out;
    }
    if (e == hipSuccess) e = hipEventSynchronize(ev[np]);
    for (size_t i = 0; i < np && e == hipSuccess; i++) e = hipEventElapsedTime(&ms_per_pass[i], ev[i], ev[i + 1]);
    for (auto &x : ev) (void) hipEventDestroy(x);
    return (int) e;
} NTT_GUARD_END

int ntt_inverse(ntt_plan_t pl, const void *d_in, void *d_out, size_t batch, int in_layout, int scale,
                void *stream) NTT_GUARD {
    int rc = check_io(pl, d_in, d_out, batch);
    if (rc) return rc;
    if (!pl->has_table) return NTT_E_NOTABLE;
    if (!pl->has_inv) return NTT_E_NOTINVERTIBLE;
    if (in_layout != NTT_LAYOUT_NATURAL && in_layout != NTT_LAYOUT_AIE_BLOCK16) return NTT_E_ARG;
    if (in_layout == NTT_LAYOUT_AIE_BLOCK16 && pl->logn < 4) return NTT_E_LAYOUT;
    if (batch == 0) return NTT_OK;
    DeviceGuard g(pl->device);
    if (g.err != hipSuccess) return (int) g.err;
    return run_inverse(pl, d_in, d_out, batch, in_layout, scale, (hipStream_t) stream);
} NTT_GUARD_END

int ntt_pointwise_mul(ntt_plan_t pl, const void *d_a, const void *d_b, void *d_out, size_t batch,
                      uint64_t scale, void *stream) NTT_GUARD {
    int rc = check_io(pl, d_a, d_b, batch);
    if (rc) return rc;
    if (batch && (!d_out || ((uintptr_t) d_out & 15u))) return NTT_E_ARG;
    if (scale >= pl->p) return NTT_E_RANGE;
    if (batch == 0) return NTT_OK;
    DeviceGuard g(pl->device);
    if (g.err != hipSuccess) return (int) g.err;
    const size_t count = batch << pl->logn;
    hipError_t e = pl->fk == FK_GL    ? ntt::launch_pointwise_gl(d_a, d_b, d_out, count, scale, (hipStream_t) stream)
                   : pl->fk == FK_M64 ? ntt::launch_pointwise_m64(d_a, d_b, d_out, count, pl->p, pl->pinv64, pl->r2_64, scale,
                                                                  (hipStream_t) stream)
                                      : ntt::launch_pointwise_m32(d_a, d_b, d_out, count, (uint32_t) pl->p, pl->pinv,
                                                                  pl->r2, (uint32_t) scale, (hipStream_t) stream);
    return (int) e;
} NTT_GUARD_END

int ntt_polymul_negacyclic(ntt_plan_t pl, void *d_a, void *d_b, void *d_out, size_t batch, void *stream) NTT_GUARD {
    int rc = check_io(pl, d_a, d_b, batch);
    if (rc) return rc;
    if (batch && (!d_out || ((uintptr_t) d_out & 15u))) return NTT_E_ARG;
    if (!pl->has_table) return NTT_E_NOTABLE;
    if (!pl->has_inv) return NTT_E_NOTINVERTIBLE;
    if (batch == 0) return NTT_OK;
    DeviceGuard g(pl->device);
    if (g.err != hipSuccess) return (int) g.err;
    hipStream_t s = (hipStream_t) stream;
    // With the Longa-Naehrig psi^-1 table the forward negacyclic NTT is the UNSCALED
    // inverse network and the inverse negacyclic NTT is N^-1 * forward network
    // (SURVEY F6-ii), so  c = Fwd( InvU(a) . InvU(b) . N^-1 ).
    const size_t operand_bytes = (batch << pl->logn) * (size_t) pl->word_bytes;
    const bool contiguous = (const char *) d_b == (const char *) d_a + operand_bytes && 2 * batch <= 0x7FFFFFFFull;
    const std::vector<PassDesc> &passes = passes_for(pl, batch);
    const PassDesc &first = passes.front();
    // Goldilocks, first (or only) pass of 7..12 stages: the radix-8 product kernel exists for that unit size.  A single-pass
    // size (2^7 <= N <= 2^12) is then ONE launch for the whole product: read a, read b, write c.
    // ... 4-byte words: radix-16 product kernel, unit sizes 2^6 .. 2^13 (any odd p: all three butterfly streams).
    bool fused_mid = first.contig && (pl->fk == FK_GL    ? ntt::have_gl_product_mid(first.log_m)
                                      : pl->fk == FK_M64 ? ntt::have_m64_product_mid(first.log_m)
                                                         : ntt::have_m32_product_mid(first.log_m));
    if (fused_mid) {
        // the product launch is not sliced: beyond blockIdx.y's range (tens of millions of tiny polynomials) take the
        // separate passes, whose launcher slices the batch.  The check IS the launcher's geometry call (product_fits).
        fused_mid = pl->fk == FK_GL    ? ntt::gl_product_mid_fits(first.log_m, pl->logn, (uint32_t) batch, pl->target_wgs)
                    : pl->fk == FK_M64 ? ntt::m64_product_mid_fits(first.log_m, pl->logn, (uint32_t) batch, pl->target_wgs)
                                       : ntt::m32_product_mid_fits(first.log_m, pl->logn, (uint32_t) batch, pl->target_wgs);
    }
    if (fused_mid) {
        // The column passes (if any) of both unscaled inverse transforms, then ONE launch that runs
        // the last inverse pass of a and of b, the pointwise product * N^-1 and the first forward pass on each
        // 2^log_m-word unit while it is workgroup-resident (3 N words of HBM traffic instead of 7 N), then the
        // forward column passes.
        for (size_t i = passes.size(); i-- > 1;) {
            const PassDesc &pd = passes[i];
            RoctxRange pass("product: inv pass", pd.contig, pd.s0, pd.log_m);
            for (int op = 0; op < (contiguous ? 1 : 2); op++) {
                void *buf = op == 0 ? d_a : d_b;
                ntt::ErasedArgs a = base_args(pl, pd, buf, buf, contiguous ? 2 * batch : batch);
                a.tw = pl->d_tw_inv;
                a.layout = NTT_LAYOUT_NATURAL;
                hipError_t e = launch_inv(pl, pd, a, s);
                if (e != hipSuccess) return (int) e;
            }
        }
        {
            RoctxRange pass("product: fused middle", 1, 0, first.log_m);
            ntt::ErasedArgs a = base_args(pl, first, d_a, d_out, batch);
            a.in2 = d_b;
            a.tw = pl->d_tw_inv;
            a.tw2 = pl->d_tw_fwd;
            a.layout = NTT_LAYOUT_NATURAL;
            a.pw_scale = to_table_form(to_table_form(pl->ninv_plain % pl->p, pl->p, pl->word_bytes), pl->p, pl->word_bytes);
            hipError_t e = pl->fk == FK_GL    ? ntt::launch_gl_product_mid(first.log_m, a, s)
                           : pl->fk == FK_M64 ? ntt::launch_m64_product_mid(first.log_m, a, s)
                                              : ntt::launch_m32_product_mid(first.log_m, a, s);
            if (e != hipSuccess) return (int) e;
        }
        for (size_t i = 1; i < passes.size(); i++) {
            const PassDesc &pd = passes[i];
            RoctxRange pass("product: fwd pass", pd.contig, pd.s0, pd.log_m);
            ntt::ErasedArgs a = base_args(pl, pd, d_out, d_out, batch);
            a.tw = pl->d_tw_fwd;
            a.layout = NTT_LAYOUT_NATURAL;
            hipError_t e = launch_fwd(pl, pd, a, s);
            if (e != hipSuccess) return (int) e;
        }
        return NTT_OK;
    }
    if (contiguous) {
        // the operands are one [2*batch][N] buffer: both unscaled inverse transforms as ONE launch per pass
        // (the decomposition is the one selected for `batch`, as ntt_plan_select documents -- not for the 2*batch rows of this launch)
        rc = run_inverse(pl, d_a, d_a, 2 * batch, NTT_LAYOUT_NATURAL, 0, s, &passes);
        if (rc) return rc;
    } else {
        rc = run_inverse(pl, d_a, d_a, batch, NTT_LAYOUT_NATURAL, 0, s, &passes);
        if (rc) return rc;
        rc = run_inverse(pl, d_b, d_b, batch, NTT_LAYOUT_NATURAL, 0, s, &passes);
        if (rc) return rc;
    }
    // pointwise product * N^-1 folded into the first pass of the final forward transform
    return run_forward(pl, d_a, d_out, batch, NTT_LAYOUT_NATURAL, s, d_b, pl->ninv_plain, &passes);
} NTT_GUARD_END

int ntt_count_noncanonical(ntt_plan_t pl, const void *d_buf, size_t batch, uint64_t *host_count) NTT_GUARD {
    if (!pl || !host_count) return NTT_E_ARG;
    *host_count = 0;
    if (batch == 0) return NTT_OK;
    if (!d_buf) return NTT_E_ARG;
    DeviceGuard g(pl->device);
    if (g.err != hipSuccess) return (int) g.err;
    std::lock_guard<std::mutex> lock(pl->counter_mu);  // the plan's one counter word
    unsigned long long *d_cnt = pl->d_counter;
    hipError_t e = hipMemset(d_cnt, 0, sizeof(*d_cnt));
    if (e == hipSuccess) e = ntt::launch_count_noncanonical(d_buf, batch << pl->logn, pl->word_bytes, pl->p, d_cnt, nullptr);
    unsigned long long h = 0;
    if (e == hipSuccess) e = hipMemcpy(&h, d_cnt, sizeof(h), hipMemcpyDeviceToHost);
    *host_count = h;
    return (int) e;
} NTT_GUARD_END

int ntt_forward_stages(ntt_plan_t pl, const void *d_in, void *d_out, size_t batch, int stage, void *stream) NTT_GUARD {
    int rc = check_io(pl, d_in, d_out, batch);
    if (rc) return rc;
    if (!pl->has_table) return NTT_E_NOTABLE;
    if (stage < 0 || stage >= pl->logn) return NTT_E_ARG;
    if (batch == 0) return NTT_OK;
    DeviceGuard g(pl->device);
    if (g.err != hipSuccess) return (int) g.err;
    hipStream_t s = (hipStream_t) stream;
    if (d_in != d_out) {
        hipError_t e = hipMemcpyAsync(d_out, d_in, (batch << pl->logn) * pl->word_bytes,
                                      hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return (int) e;
    }
    for (int st = 0; st <= stage; st++) {
        hipError_t e = pl->fk == FK_GL    ? ntt::launch_stage_gl(d_out, pl->d_tw_fwd, pl->logn, st, batch, s)
                       : pl->fk == FK_M64 ? ntt::launch_stage_m64(d_out, pl->d_tw_fwd, pl->logn, st, batch, pl->p, pl->pinv64, pl->r2_64, s)
                                          : ntt::launch_stage_m32(d_out, pl->d_tw_fwd, pl->logn, st, batch,
                                                                  (uint32_t) pl->p, pl->pinv, pl->r2, s);
        if (e != hipSuccess) return (int) e;
    }
    return NTT_OK;
} NTT_GUARD_END

}  // extern "C"
