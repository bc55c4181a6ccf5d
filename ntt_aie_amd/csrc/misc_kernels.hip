// misc_kernels.hip -- the small kernels around the pass kernels:
//   * pointwise product (config 4's middle leg; no reference counterpart),
//   * one-stage-per-launch network (the reference's test_stage hook,
//     src/test.cpp:55-58, 67: "run stages 0..stage and stop"), one thread per
//     butterfly exactly as src/test.cpp:42-51 enumerates them.  Bring-up path
//     and an independent second GPU implementation for the parity tests.
#include <hip/hip_runtime.h>

#include "field.h"
#include "kernels.h"

#ifndef NTT_PW_UNROLL
#define NTT_PW_UNROLL 2
#endif
#ifndef NTT_PW_NT
#define NTT_PW_NT 3  // bit 0: non-temporal loads, bit 1: non-temporal stores
#endif

namespace ntt {
namespace {

template <class W, int V>
struct alignas(sizeof(W) * V) Vec {
    W v[V];
};

template <class F>
__global__ __launch_bounds__(256) void pointwise_kernel(const typename F::W *a, const typename F::W *b,
                                                        typename F::W *c, size_t count, F f,
                                                        typename F::W scale, int use_scale) {
    using W = typename F::W;
    constexpr int V = 16 / sizeof(W);
    using Ch = Vec<W, V>;
    const size_t nchunks = count / V;  // count is a multiple of N >= 2; tail handled below
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    // every word is read once and written once: non-temporal accesses, NTT_PW_UNROLL chunks of each operand requested before the
    // first product (same-process A/B in profiles/r02_pointwise.txt)
    using u32x4 = unsigned int __attribute__((ext_vector_type(4)));
    const u32x4 *pa = reinterpret_cast<const u32x4 *>(a), *pb = reinterpret_cast<const u32x4 *>(b);
    u32x4 *pc = reinterpret_cast<u32x4 *>(c);
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (NTT_PW_UNROLL - 1) * stride < nchunks; i += NTT_PW_UNROLL * stride) {
        u32x4 xs[NTT_PW_UNROLL], ys[NTT_PW_UNROLL];
#pragma unroll
        for (int u = 0; u < NTT_PW_UNROLL; ++u) {
            xs[u] = (NTT_PW_NT & 1) ? __builtin_nontemporal_load(pa + i + u * stride) : pa[i + u * stride];
            ys[u] = (NTT_PW_NT & 1) ? __builtin_nontemporal_load(pb + i + u * stride) : pb[i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < NTT_PW_UNROLL; ++u) {
            Ch x, y, z;
            __builtin_memcpy(&x, &xs[u], 16);
            __builtin_memcpy(&y, &ys[u], 16);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                W t = f.mul_plain(x.v[k], y.v[k]);
                z.v[k] = use_scale ? f.mul_plain(t, scale) : t;
            }
            u32x4 zz;
            __builtin_memcpy(&zz, &z, 16);
            if (NTT_PW_NT & 2) __builtin_nontemporal_store(zz, pc + i + u * stride);
            else pc[i + u * stride] = zz;
        }
    }
    for (; i < nchunks; i += stride) {
        Ch x = reinterpret_cast<const Ch *>(a)[i];
        Ch y = reinterpret_cast<const Ch *>(b)[i];
        Ch z;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            W t = f.mul_plain(x.v[k], y.v[k]);
            z.v[k] = use_scale ? f.mul_plain(t, scale) : t;
        }
        reinterpret_cast<Ch *>(c)[i] = z;
    }
    // tail (count not a multiple of V: only N = 2 with 4-byte words and odd batch)
    for (size_t i = nchunks * V + (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        W t = f.mul_plain(a[i], b[i]);
        c[i] = use_scale ? f.mul_plain(t, scale) : t;
    }
}

// stage s: t = 2^s, butterfly k of a polynomial: block i = k >> s, j = (i << (s+1)) | (k & (t-1))
template <class F>
__global__ __launch_bounds__(256) void stage_kernel(typename F::W *data, const typename F::W *tw, int n,
                                                    int s, size_t total_bf, F f) {
    using W = typename F::W;
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    const int half = n - 1;
    for (size_t g = (size_t) blockIdx.x * blockDim.x + threadIdx.x; g < total_bf; g += stride) {
        const size_t poly = g >> half;
        const uint32_t k = (uint32_t) (g & ((1u << half) - 1u));
        const uint32_t t = 1u << s;
        const uint32_t i = k >> s;
        const uint32_t j = (i << (s + 1)) | (k & (t - 1u));
        const uint32_t h = 1u << (n - s - 1);
        W *a = data + (poly << n);
        const W root = tw[h + i];  // table form
        const W v0 = a[j], v1 = a[j + t];
        a[j] = f.add(v0, v1);
        a[j + t] = f.mul(f.sub(v0, v1), root);
    }
}

// T[i] = base^e(i) in table (Montgomery) form, e(i) by the table rule (plan.h:make_table):
//   kind 0: e = i                         (src/test.cpp:27-32: natural-order powers)
//   kind 1: e = bitrev_{log2 h}(i - h) * N/(2h), h = 2^floor(log2 i)      (cyclic)
//   kind 2: e = bitrev_{logN}(i)                                          (negacyclic, base = psi^-1)
// Square-and-multiply per entry: log2(N) products, the whole table in one launch -- no host upload.
template <class F>
__global__ __launch_bounds__(256) void gen_table_kernel(typename F::W *T, int logn, int kind,
                                                        typename F::W base_m, typename F::W one_m, F f) {
    using W = typename F::W;
    const uint32_t N = 1u << logn;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        uint32_t e = i;
        if (kind == 1) {
            if (i == 0) {
                e = 0;
            } else {
                const int lh = 31 - __clz(i);
                const uint32_t idx = i - (1u << lh);
                const uint32_t rev = lh ? (__brev(idx) >> (32 - lh)) : 0u;
                e = rev << (logn - lh - 1);
            }
        } else if (kind == 2) {
            e = __brev(i) >> (32 - logn);
        }
        W r = one_m, b = base_m;
        while (e) {
            if (e & 1u) r = f.mul(r, b);
            b = f.mul(b, b);
            e >>= 1;
        }
        T[i] = r;
    }
}

// out[i] = T[i] * c, both in table (Montgomery) form: the scaled stage-0 twiddles of the inverse transform (pass.h: fold_scale)
template <class F>
__global__ __launch_bounds__(256) void scale_table_kernel(const typename F::W *T, typename F::W *out, size_t count,
                                                          typename F::W c_m, F f) {
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t) gridDim.x * blockDim.x)
        out[i] = f.mul(T[i], c_m);
}

// precondition check: number of words >= p (the kernels, like the reference's vector_modadd /
// vector_modsub, src/aie_core.cc:41-62, assume canonical residues)
template <class W>
__global__ __launch_bounds__(256) void count_noncanonical_kernel(const W *a, size_t count, W p, unsigned long long *out) {
    unsigned long long bad = 0;
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) bad += a[i] >= p;
    for (int off = 32; off; off >>= 1) bad += __shfl_down(bad, off, 64);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(out, bad);
}

inline unsigned grid_for(size_t work) {
    size_t g = (work + 255) / 256;
    if (g > 8192) g = 8192;  // 256 CUs x 8 x 4: grid-stride the rest
    if (g == 0) g = 1;
    return (unsigned) g;
}

}  // namespace

hipError_t launch_pointwise_gl(const void *a, const void *b, void *c, size_t count, uint64_t scale,
                               hipStream_t s) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(pointwise_kernel<FieldGL>, dim3(grid_for(count / 2)), dim3(256), 0, s,
                       (const uint64_t *) a, (const uint64_t *) b, (uint64_t *) c, count, FieldGL{}, scale,
                       scale != 1 ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_pointwise_m32(const void *a, const void *b, void *c, size_t count, uint32_t p,
                                uint32_t pinv, uint32_t r2, uint32_t scale, hipStream_t s) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(pointwise_kernel<FieldM32>, dim3(grid_for(count / 4)), dim3(256), 0, s,
                       (const uint32_t *) a, (const uint32_t *) b, (uint32_t *) c, count,
                       FieldM32{p, pinv, r2}, scale, scale != 1 ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_pointwise_m64(const void *a, const void *b, void *c, size_t count, uint64_t p,
                                uint64_t pinv, uint64_t r2, uint64_t scale, hipStream_t s) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(pointwise_kernel<FieldM64>, dim3(grid_for(count / 2)), dim3(256), 0, s,
                       (const uint64_t *) a, (const uint64_t *) b, (uint64_t *) c, count,
                       FieldM64{p, pinv, r2}, scale, scale != 1 ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_gen_table_gl(void *T, int logn, int kind, uint64_t base_m, uint64_t one_m, hipStream_t s) {
    hipLaunchKernelGGL(gen_table_kernel<FieldGL>, dim3(grid_for((size_t) 1 << logn)), dim3(256), 0, s,
                       (uint64_t *) T, logn, kind, base_m, one_m, FieldGL{});
    return hipGetLastError();
}

hipError_t launch_gen_table_m32(void *T, int logn, int kind, uint32_t base_m, uint32_t one_m, uint32_t p,
                                uint32_t pinv, uint32_t r2, hipStream_t s) {
    hipLaunchKernelGGL(gen_table_kernel<FieldM32>, dim3(grid_for((size_t) 1 << logn)), dim3(256), 0, s,
                       (uint32_t *) T, logn, kind, base_m, one_m, FieldM32{p, pinv, r2});
    return hipGetLastError();
}

hipError_t launch_gen_table_m64(void *T, int logn, int kind, uint64_t base_m, uint64_t one_m, uint64_t p,
                                uint64_t pinv, uint64_t r2, hipStream_t s) {
    hipLaunchKernelGGL(gen_table_kernel<FieldM64>, dim3(grid_for((size_t) 1 << logn)), dim3(256), 0, s,
                       (uint64_t *) T, logn, kind, base_m, one_m, FieldM64{p, pinv, r2});
    return hipGetLastError();
}

hipError_t launch_scale_table_gl(const void *T, void *out, size_t count, uint64_t c_m, hipStream_t s) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(scale_table_kernel<FieldGL>, dim3(grid_for(count)), dim3(256), 0, s, (const uint64_t *) T,
                       (uint64_t *) out, count, c_m, FieldGL{});
    return hipGetLastError();
}

hipError_t launch_scale_table_m64(const void *T, void *out, size_t count, uint64_t c_m, uint64_t p, uint64_t pinv, uint64_t r2,
                                  hipStream_t s) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(scale_table_kernel<FieldM64>, dim3(grid_for(count)), dim3(256), 0, s, (const uint64_t *) T,
                       (uint64_t *) out, count, c_m, FieldM64{p, pinv, r2});
    return hipGetLastError();
}

hipError_t launch_count_noncanonical(const void *a, size_t count, int word_bytes, uint64_t p, void *d_out, hipStream_t s) {
    if (word_bytes == 8)
        hipLaunchKernelGGL(count_noncanonical_kernel<uint64_t>, dim3(grid_for(count)), dim3(256), 0, s,
                           (const uint64_t *) a, count, p, (unsigned long long *) d_out);
    else
        hipLaunchKernelGGL(count_noncanonical_kernel<uint32_t>, dim3(grid_for(count)), dim3(256), 0, s,
                           (const uint32_t *) a, count, (uint32_t) p, (unsigned long long *) d_out);
    return hipGetLastError();
}

hipError_t launch_stage_gl(void *data, const void *tw, int n, int stage, size_t batch, hipStream_t s) {
    const size_t total = batch << (n - 1);
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(stage_kernel<FieldGL>, dim3(grid_for(total)), dim3(256), 0, s, (uint64_t *) data,
                       (const uint64_t *) tw, n, stage, total, FieldGL{});
    return hipGetLastError();
}

hipError_t launch_stage_m32(void *data, const void *tw, int n, int stage, size_t batch, uint32_t p,
                            uint32_t pinv, uint32_t r2, hipStream_t s) {
    const size_t total = batch << (n - 1);
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(stage_kernel<FieldM32>, dim3(grid_for(total)), dim3(256), 0, s, (uint32_t *) data,
                       (const uint32_t *) tw, n, stage, total, FieldM32{p, pinv, r2});
    return hipGetLastError();
}

hipError_t launch_stage_m64(void *data, const void *tw, int n, int stage, size_t batch, uint64_t p,
                            uint64_t pinv, uint64_t r2, hipStream_t s) {
    const size_t total = batch << (n - 1);
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(stage_kernel<FieldM64>, dim3(grid_for(total)), dim3(256), 0, s, (uint64_t *) data,
                       (const uint64_t *) tw, n, stage, total, FieldM64{p, pinv, r2});
    return hipGetLastError();
}

}  // namespace ntt
