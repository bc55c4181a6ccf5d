// copy_bw.hip -- achievable read+write stream rate on this device for the access widths the pass
// kernels use (8 B and 16 B per lane) and for read-only / write-only streams.  2 GiB in, 2 GiB out.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <class V> __global__ void k_copy(const V *in, V *out, size_t n) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x, s = (size_t) gridDim.x * blockDim.x;
    for (; i < n; i += s) out[i] = in[i];
}
template <class V> __global__ void k_read(const V *in, V *out, size_t n) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x, s = (size_t) gridDim.x * blockDim.x;
    V acc = in[i];
    for (i += s; i < n; i += s) { V v = in[i]; acc.x ^= v.x; acc.y ^= v.y; }
    if (acc.x == 0x12345678u) out[0] = acc;
}
template <class V> __global__ void k_write(V *out, size_t n, V v) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x, s = (size_t) gridDim.x * blockDim.x;
    for (; i < n; i += s) out[i] = v;
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
    for (int grid : {2048, 8192, 32768}) {
        double t;
        t = timeit([&] { hipLaunchKernelGGL(k_copy<uint2>, dim3(grid), dim3(256), 0, 0, (const uint2 *) a, (uint2 *) b, bytes / 8); });
        printf("grid %5d copy  8B/lane: %.2f TB/s (read+write)\n", grid, 2.0 * bytes / t / 1e12);
        t = timeit([&] { hipLaunchKernelGGL(k_copy<uint4>, dim3(grid), dim3(256), 0, 0, (const uint4 *) a, (uint4 *) b, bytes / 16); });
        printf("grid %5d copy 16B/lane: %.2f TB/s (read+write)\n", grid, 2.0 * bytes / t / 1e12);
        t = timeit([&] { hipLaunchKernelGGL(k_read<uint4>, dim3(grid), dim3(256), 0, 0, (const uint4 *) a, (uint4 *) b, bytes / 16); });
        printf("grid %5d read  16B/lane: %.2f TB/s\n", grid, 1.0 * bytes / t / 1e12);
        t = timeit([&] { hipLaunchKernelGGL(k_write<uint4>, dim3(grid), dim3(256), 0, 0, (uint4 *) b, bytes / 16, make_uint4(1, 2, 3, 4)); });
        printf("grid %5d write 16B/lane: %.2f TB/s\n", grid, 1.0 * bytes / t / 1e12);
        t = timeit([&] { hipLaunchKernelGGL(k_write<uint2>, dim3(grid), dim3(256), 0, 0, (uint2 *) b, bytes / 8, make_uint2(1, 2)); });
        printf("grid %5d write  8B/lane: %.2f TB/s\n", grid, 1.0 * bytes / t / 1e12);
    }
    return 0;
}
