// wg_probe.hip -- where do the workgroups of a one-generation launch land, and what does HW_REG_LDS_ALLOC say about the
// workgroups that share a CU?  1024 workgroups x 256 threads x 17408 B of LDS (the shape of BASELINE config 2's launch).
// Prints, per workgroup: XCC, SE, CU, LDS_ALLOC raw, start time; then the co-residents of a few CUs.
// build: hipcc --offload-arch=gfx950 -O2 tools/wg_probe.hip -o tools/wg_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#include <map>
#include <algorithm>

struct Rec { uint32_t hw_id, lds_alloc, xcc; uint64_t t0, t1; };

__global__ __launch_bounds__(256) void probe(Rec *out, int spin) {
    extern __shared__ uint32_t tile[];
    uint32_t hw, la, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(la));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const uint64_t t0 = __builtin_readcyclecounter();
    tile[threadIdx.x] = threadIdx.x;
    __syncthreads();
    uint32_t acc = tile[(threadIdx.x * 7) & 255];
    for (int i = 0; i < spin; ++i) acc = acc * 1664525u + 1013904223u;
    tile[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        Rec r{hw, la, xcc, t0, (uint64_t) __builtin_readcyclecounter()};
        r.hw_id ^= (tile[5] & 0);  // keep the loop
        out[blockIdx.x] = r;
    }
}

int main(int argc, char **argv) {
    const int WGS = argc > 2 ? atoi(argv[2]) : 1024;
    const int LDS = argc > 1 ? atoi(argv[1]) : 17408;  // bytes of LDS per workgroup
    Rec *d;
    hipMalloc(&d, WGS * sizeof(Rec));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe, dim3(WGS), dim3(256), LDS, 0, d, 20000);
        hipDeviceSynchronize();
    }
    std::vector<Rec> h(WGS);
    hipMemcpy(h.data(), d, WGS * sizeof(Rec), hipMemcpyDeviceToHost);
    uint64_t tmin = ~0ull;
    for (auto &r : h) tmin = std::min(tmin, r.t0);
    std::map<uint32_t, std::vector<int>> by_cu;
    for (int i = 0; i < WGS; ++i) {
        const Rec &r = h[i];
        const uint32_t cu = (r.hw_id >> 8) & 0xf, sh = (r.hw_id >> 12) & 1, se = (r.hw_id >> 13) & 7, xcc = r.xcc & 0xf;
        by_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(i);
        if (i < 48) printf("wg %4d  hw_id %08x xcc %u se %u sh %u cu %2u  lds_alloc %08x  t0 %8llu t1 %8llu\n", i, r.hw_id, xcc, se, sh, cu, r.lds_alloc,
                           (unsigned long long) (r.t0 - tmin), (unsigned long long) (r.t1 - tmin));
    }
    printf("distinct CUs: %zu\n", by_cu.size());
    int shown = 0;
    std::map<size_t, int> hist;
    for (auto &kv : by_cu) {
        hist[kv.second.size()]++;
        if (shown++ < 6) {
            printf("cu key %05x:", kv.first);
            for (int i : kv.second) printf("  wg %d (lds %08x, t0 %llu)", i, h[i].lds_alloc, (unsigned long long) (h[i].t0 - tmin));
            printf("\n");
        }
    }
    for (auto &kv : hist) printf("%d CUs hold %zu workgroups\n", kv.second, kv.first);
    std::map<int, int> conc;  // workgroups of a CU that started before the first one of that CU ended
    for (auto &kv : by_cu) {
        uint64_t first_end = ~0ull;
        for (int i : kv.second) first_end = std::min(first_end, h[i].t1);
        int n = 0;
        for (int i : kv.second) n += h[i].t0 < first_end;
        conc[n]++;
    }
    for (auto &kv : conc) printf("LDS %d B per workgroup: %d CUs ran %d workgroups at once\n", LDS, kv.second, kv.first);
    return 0;
}
