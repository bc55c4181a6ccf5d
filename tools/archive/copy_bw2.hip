// copy_bw2.hip -- copy kernels with several loads in flight per lane (upper bound of the r+w stream rate)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
template <int U> __global__ __launch_bounds__(256) void k_copy(const v4u *in, v4u *out, size_t n) {
    size_t i = ((size_t) blockIdx.x * blockDim.x) * U + threadIdx.x, s = (size_t) gridDim.x * blockDim.x * U;
    for (; i + (U - 1) * 256 < n; i += s) {
        v4u v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(&in[i + u * 256]);
#pragma unroll
        for (int u = 0; u < U; u++) __builtin_nontemporal_store(v[u], &out[i + u * 256]);
    }
}
template <class F> double timeit(F f) {
    hipEvent_t a, b; (void) hipEventCreate(&a); (void) hipEventCreate(&b);
    f(); f();
    (void) hipEventRecord(a, 0); for (int r = 0; r < 10; r++) f(); (void) hipEventRecord(b, 0);
    (void) hipDeviceSynchronize(); float ms; (void) hipEventElapsedTime(&ms, a, b); return ms / 10 * 1e-3;
}
int main() {
    const size_t bytes = 2ull << 30;
    void *a, *b; if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
    (void) hipMemset(a, 1, bytes); (void) hipMemset(b, 2, bytes);
    for (int grid : {1024, 2048, 4096, 8192, 16384}) {
        double t1 = timeit([&] { hipLaunchKernelGGL(k_copy<1>, dim3(grid), dim3(256), 0, 0, (const v4u *) a, (v4u *) b, bytes / 16); });
        double t4 = timeit([&] { hipLaunchKernelGGL(k_copy<4>, dim3(grid), dim3(256), 0, 0, (const v4u *) a, (v4u *) b, bytes / 16); });
        double t8 = timeit([&] { hipLaunchKernelGGL(k_copy<8>, dim3(grid), dim3(256), 0, 0, (const v4u *) a, (v4u *) b, bytes / 16); });
        printf("grid %5d nt copy 16B/lane: unroll1 %.2f  unroll4 %.2f  unroll8 %.2f TB/s (read+write)\n", grid,
               2.0 * bytes / t1 / 1e12, 2.0 * bytes / t4 / 1e12, 2.0 * bytes / t8 / 1e12);
    }
    return 0;
}
