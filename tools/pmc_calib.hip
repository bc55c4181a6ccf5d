// pmc_calib.hip -- known-byte-count streams to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on
// gfx950 for the two access widths the pass kernels use (8 B and 16 B per lane).
// Each kernel reads 1 GiB and writes 1 GiB.  Run under:
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./pmc_calib
//   rocprofv3 --pmc WRITE_SIZE --output-format csv -d out -- ./pmc_calib
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void calib_copy_8B(const uint2 *in, uint2 *out, size_t n) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x, s = (size_t) gridDim.x * blockDim.x;
    for (; i < n; i += s) out[i] = in[i];
}
__global__ void calib_copy_16B(const uint4 *in, uint4 *out, size_t n) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x, s = (size_t) gridDim.x * blockDim.x;
    for (; i < n; i += s) out[i] = in[i];
}
int main() {
    const size_t bytes = 1ull << 30;
    void *a, *b;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
    (void) hipMemset(a, 1, bytes);
    (void) hipMemset(b, 2, bytes);
    for (int r = 0; r < 3; r++) {
        hipLaunchKernelGGL(calib_copy_8B, dim3(8192), dim3(256), 0, 0, (const uint2 *) a, (uint2 *) b, bytes / 8);
        hipLaunchKernelGGL(calib_copy_16B, dim3(8192), dim3(256), 0, 0, (const uint4 *) a, (uint4 *) b, bytes / 16);
    }
    (void) hipDeviceSynchronize();
    printf("calib done: each launch read %zu B and wrote %zu B\n", bytes, bytes);
    return 0;
}
