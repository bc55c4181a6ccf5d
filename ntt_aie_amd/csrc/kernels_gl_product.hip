// kernels_gl_product.hip -- the fused middle pass of the Goldilocks negacyclic product (pass.h: run_product_pass):
// last inverse-network pass of both operands + pointwise product + first forward-network pass in one workgroup-resident
// sweep over each 2^LOG_M-word unit (no reference counterpart: the reference has no product).
#include <hip/hip_runtime.h>
#include <string.h>

#include "kernels.h"
#include "pass.h"

namespace ntt {
namespace {

template <class CI, class CF>
struct GpuProductExec {
    using W = typename CI::W;
    Ctx<CI> ci;
    Ctx<CF> cf;
    W keep[CI::E];  // the transformed words of operand a while operand b is transformed
    W pre[CI::E];   // register prefetch: operand b of this unit / operand a of the next one
    W *tile, *tab_i, *tab_f;
    __device__ __forceinline__ void init(const PassArgs<CI> &aa, const PassArgs<CF> &af) {
        phase_init<CI>(ci, aa, threadIdx.x, blockIdx.x, blockIdx.y);
        phase_init<CF>(cf, af, threadIdx.x, blockIdx.x, blockIdx.y);
    }
    template <class Fn>
    __device__ __forceinline__ void eachI(Fn &&f) { f(ci); }
    template <class Fn>
    __device__ __forceinline__ void eachF(Fn &&f) { f(cf); }
    template <class Fn>
    __device__ __forceinline__ void eachIF(Fn &&f) { f(ci, cf, keep, pre); }
    __device__ __forceinline__ void sync(std::false_type) { __syncthreads(); }
    __device__ __forceinline__ void sync(std::true_type) {  // wave-local unit: LDS operations of one wave execute in order
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __device__ __forceinline__ uint32_t pg_base() const { return ci.pg_base; }
    __device__ __forceinline__ W *lds() { return tile; }
    __device__ __forceinline__ W *tabI() { return tab_i; }
    __device__ __forceinline__ W *tabF() { return tab_f; }
};

#ifndef NTT_PRODUCT_WPE
#define NTT_PRODUCT_WPE 4  // waves per SIMD the register allocator must leave room for (128 VGPRs)
#endif
template <class CI, class CF>
__global__ __launch_bounds__(CI::NT, NTT_PRODUCT_WPE)
void product_kernel(PassArgs<CI> aa, const typename CI::W *in_b, PassArgs<CF> af) {
    __shared__ __attribute__((aligned(16))) typename CI::W tile[CI::LDS_WORDS];
    __shared__ __attribute__((aligned(16))) typename CI::W tab_i[tw_table_words<CI>()];
    __shared__ __attribute__((aligned(16))) typename CI::W tab_f[tw_table_words<CF>()];
    GpuProductExec<CI, CF> ex;
    ex.tile = tile;
    ex.tab_i = tab_i;
    ex.tab_f = tab_f;
    PassArgs<CI> ab = aa;
    ab.in = in_b;
    run_product_pass<CI, CF>(ex, aa, ab, af);
}

template <int LOG_M>
hipError_t launch_product(const ErasedArgs &e, hipStream_t s) {
    using CI = typename ProductCfg<LOG_M>::CI;
    using CF = typename ProductCfg<LOG_M>::CF;
    using W = uint64_t;
    PassGeom g = pass_geometry(e.n, 0, LOG_M, 0, CI::LOG_U, true, e.batch, e.target_wgs);
    if (g.grid_y == 0) return hipSuccess;
    if (g.grid_y > 65535u) return hipErrorInvalidValue;  // callers fall back to the separate passes
    PassArgs<CI> aa;
    ::memset((void *) &aa, 0, sizeof(aa));
    aa.in = (const W *) e.in;
    aa.out = nullptr;
    aa.tw = (const W *) e.tw;
    aa.n = e.n;
    aa.s0 = 0;
    aa.batch = e.batch;
    aa.ppw = g.ppw;
    aa.log_ul = g.log_ul;
    aa.log_uh = g.log_uh;
    aa.log_up = g.log_up;
    aa.layout = LAYOUT_NATURAL;
    aa.pg_stride = 1;
    PassArgs<CF> af;
    ::memset((void *) &af, 0, sizeof(af));
    af.in = nullptr;
    af.out = (W *) e.out;
    af.tw = (const W *) e.tw2;
    af.n = e.n;
    af.s0 = 0;
    af.batch = e.batch;
    af.ppw = g.ppw;
    af.log_ul = g.log_ul;
    af.log_uh = g.log_uh;
    af.log_up = g.log_up;
    af.layout = e.layout;
    af.pg_stride = 1;
    af.pw_scale = (W) e.pw_scale;
    hipLaunchKernelGGL((product_kernel<CI, CF>), dim3(g.grid_x, g.grid_y, 1), dim3(CI::NT, 1, 1), 0, s, aa, (const W *) e.in2, af);
    return hipGetLastError();
}

}  // namespace

bool have_gl_product_mid(int log_m) { return log_m >= 7 && log_m <= 12; }

hipError_t launch_gl_product_mid(int log_m, const ErasedArgs &a, hipStream_t s) {
    switch (log_m) {
        case 7: return launch_product<7>(a, s);
        case 8: return launch_product<8>(a, s);
        case 9: return launch_product<9>(a, s);
        case 10: return launch_product<10>(a, s);
        case 11: return launch_product<11>(a, s);
        case 12: return launch_product<12>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ntt
