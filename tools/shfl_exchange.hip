// shfl_exchange.hip -- the measurement behind DESIGN.md section 7's deviation from north_star ("wavefront __shfl for the small-stride
// inner stages").  In this engine the small strides never leave a thread (they are the radix-8 register round); what is left between
// rounds is an EXCHANGE that swaps 3 register-index bits with 3 lane-index bits: word e of lane l  <->  word (l & 7) of lane
// ((l & ~7) | e), an 8 x 8 transpose inside every group of 8 lanes.  Three ways to do it, same data movement:
//   lds    : 8 ds_write_b64 + 8 ds_read_b64 per thread through the wave's own LDS segment (what csrc/pass.h does; wave-local, no barrier),
//            unpadded (the LDS-DMA tile of the headline first pass: 4-way bank conflicts) and padded (16 bytes per 8 words, the other tiles)
//   shfl   : __shfl_xor butterflies (ds_bpermute_b32: the "wavefront __shfl" of north_star): 3 steps x 4 pairs x (select, shuffle, select)
//   dpp    : the same butterflies with DPP quad_perm for the xor-1 / xor-2 steps and ds_swizzle for xor-4 (no xor-4 DPP row mode on gfx9)
// Each round is the exchange plus W "butterfly-like" 64-bit multiply-adds per word (W = 0: the exchange alone; W = 3 is about the VALU
// work of one real radix-8 round: 3 stages x 4 butterflies x 22 VALU per 8 words), all SIMDs busy, 4 waves per SIMD.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/shfl_exchange.hip -o /tmp/shfl_exchange && /tmp/shfl_exchange
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void work(uint64_t (&x)[8], int W, uint64_t c) {
    for (int w = 0; w < W; ++w) {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = x[e] * c + (x[e] >> 7);  // a 64-bit multiply-add chain: 3 multiplies + adds per word
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void exchange_kernel(uint64_t *out, int rounds, int W, uint64_t c) {
    __shared__ uint64_t tile[256 * 8 + 256 * 2];
    auto pad = [](uint32_t i) { return MODE == 3 ? i + (i >> 3) * 2 : i; };  // 16 bytes after every 8 words (csrc/pass.h: lds_index)
    const uint32_t t = threadIdx.x, lane = t & 63u;
    uint64_t x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (uint64_t) (blockIdx.x * 256u + t) * 8u + e;
    for (int r = 0; r < rounds; ++r) {
        work(x, W, c);
        if constexpr (MODE == 0 || MODE == 3) {
            // the wave's own segment: word (q, e) at 8q + e, read back at (q & ~7 | e') * 8 + (q & 7)
#pragma unroll
            for (int e = 0; e < 8; ++e) tile[pad(t * 8 + e)] = x[e];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = tile[pad(((t & ~7u) | (uint32_t) e) * 8 + (t & 7u))];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else {
#pragma unroll
            for (int k = 1; k < 8; k <<= 1) {
                const bool up = (lane & (uint32_t) k) != 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (e & k) continue;
                    const uint64_t a = x[e], b = x[e | k];
                    const uint64_t send = up ? a : b;
                    uint64_t recv;
                    if constexpr (MODE == 1) {
                        recv = __shfl_xor(send, k, 64);
                    } else {
                        uint32_t lo = (uint32_t) send, hi = (uint32_t) (send >> 32);
                        if (k == 1) {  // quad_perm [1,0,3,2]
                            lo = __builtin_amdgcn_update_dpp(0u, lo, 0xB1, 0xF, 0xF, true);
                            hi = __builtin_amdgcn_update_dpp(0u, hi, 0xB1, 0xF, 0xF, true);
                        } else if (k == 2) {  // quad_perm [2,3,0,1]
                            lo = __builtin_amdgcn_update_dpp(0u, lo, 0x4E, 0xF, 0xF, true);
                            hi = __builtin_amdgcn_update_dpp(0u, hi, 0x4E, 0xF, 0xF, true);
                        } else {  // ds_swizzle bit mode: and 0x1F, or 0, xor 4
                            lo = (uint32_t) __builtin_amdgcn_ds_swizzle((int) lo, 0x101F);
                            hi = (uint32_t) __builtin_amdgcn_ds_swizzle((int) hi, 0x101F);
                        }
                        recv = ((uint64_t) hi << 32) | lo;
                    }
                    x[e] = up ? recv : a;
                    x[e | k] = up ? b : recv;
                }
            }
        }
    }
    uint64_t acc = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc ^= x[e] + (uint64_t) e;
    out[(size_t) blockIdx.x * 256 + t] = acc;
}

int main() {
    const int blocks = 256 * 4 * 4, rounds = 400;  // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    uint64_t *d = nullptr;
    CHECK(hipMalloc(&d, (size_t) blocks * 256 * 8));
    std::vector<uint64_t> ref((size_t) blocks * 256), got(ref.size());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const char *names[4] = {"lds unpadded", "shfl (ds_bpermute)", "dpp + ds_swizzle", "lds padded"};
    printf("# 8 x 8 transpose of 8-byte words inside every group of 8 lanes (the exchange between two radix-8 rounds), %d rounds, %d workgroups x 256 threads\n", rounds, blocks);
    for (int W : {0, 1, 2, 3, 4, 6, 8}) {
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CHECK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(exchange_kernel<0>, dim3(blocks), dim3(256), 0, 0, d, rounds, W, 0x9E3779B97F4A7C15ull);
                if (mode == 1) hipLaunchKernelGGL(exchange_kernel<1>, dim3(blocks), dim3(256), 0, 0, d, rounds, W, 0x9E3779B97F4A7C15ull);
                if (mode == 2) hipLaunchKernelGGL(exchange_kernel<2>, dim3(blocks), dim3(256), 0, 0, d, rounds, W, 0x9E3779B97F4A7C15ull);
                if (mode == 3) hipLaunchKernelGGL(exchange_kernel<3>, dim3(blocks), dim3(256), 0, 0, d, rounds, W, 0x9E3779B97F4A7C15ull);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            CHECK(hipMemcpy(got.data(), d, got.size() * 8, hipMemcpyDeviceToHost));
            if (mode == 0) ref = got;
            const bool same = got == ref;
            // per wave and round: 4 waves per SIMD share the SIMD, so wall time per round of ONE wave slot = best / rounds
            printf("work %d  %-20s %8.3f ms  = %7.1f ns per round per SIMD-slot  %s\n", W, names[mode], best, best * 1e6 / rounds / 4.0, same ? "outputs identical" : "OUTPUTS DIFFER");
        }
    }
    (void) hipFree(d);
    return 0;
}
