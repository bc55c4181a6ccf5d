// dma_test.hip -- minimal check of the LDS-DMA recipe (global_load_lds_dwordx4) used by pass.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
__device__ __forceinline__ void glds16(const void *gptr, uint32_t lds_byte) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gptr), "s"(lds_byte) : "memory");
}
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint64_t *in, uint64_t *out) {
    __shared__ __attribute__((aligned(16))) uint64_t tile[4096];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t wbase = wave * 512;  // words
    const uint64_t *g = in + (size_t) blockIdx.x * 2048 + wbase + lane * 2;
    if (MODE == 0) {
        const uint32_t lds0 = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) uint64_t *) tile;
        const uint32_t wl = __builtin_amdgcn_readfirstlane(lds0 + wbase * 8);
        for (int i = 0; i < 4; i++) glds16(g + i * 128, wl + i * 1024);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        for (int i = 0; i < 4; i++)
            __builtin_amdgcn_global_load_lds(g + i * 128, (__attribute__((address_space(3))) void *) (tile + wbase + i * 128), 16, 0, 0);
        __builtin_amdgcn_s_waitcnt(0);
    }
    __syncthreads();
    for (int i = 0; i < 8; i++) out[(size_t) blockIdx.x * 2048 + i * 256 + tid] = tile[i * 256 + tid] + 1;
}
int main() {
    const int blocks = 64;
    std::vector<uint64_t> h(blocks * 2048), r(blocks * 2048);
    for (size_t i = 0; i < h.size(); i++) h[i] = i * 0x9E3779B97F4A7C15ull;
    uint64_t *din, *dout;
    hipMalloc(&din, h.size() * 8); hipMalloc(&dout, h.size() * 8);
    hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; mode++) {
        hipMemset(dout, 0, h.size() * 8);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, din, dout);
        else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, din, dout);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(r.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < h.size(); i++) bad += r[i] != h[i] + 1;
        printf("mode %d: sync=%d mismatches=%zu first: got %llx want %llx\n", mode, (int) e, bad,
               (unsigned long long) r[1], (unsigned long long) (h[1] + 1));
        fflush(stdout);
    }
    return 0;
}
