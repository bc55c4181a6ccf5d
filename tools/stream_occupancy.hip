// stream_occupancy.hip -- the forward Goldilocks butterfly stream (two butterflies per statement: csrc/gl_asm.h) as a register-resident
// radix-8 round, at 1 .. 8 waves per SIMD.  tools/valu_issue_cost.hip measures the streams only up to 4 waves (its kernels pin
// v104+); here the statement's scratch lives at v[40:63] and its carries in s[50:69] (tools/gen_gl_asm.py with NTT_GEN_W=1 ->
// ab/gl_asm_w.h: gl_fwd2_v_w), so the kernel fits 64 VGPRs and 8 waves share a SIMD.  Question (round 5): does the stream itself
// speed up with occupancy the way its single instruction forms do (3.35 -> 1.99 cycles between 4 and 8 waves)?
// build: NTT_GEN_W=1 python3 tools/gen_gl_asm.py ab/gl_asm_w.h && hipcc -O3 --offload-arch=gfx950 -I ab tools/stream_occupancy.hip -o tools/stream_occupancy
// run:   tools/stream_occupancy   (text on stdout)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "gl_asm_w.h"

constexpr int ITERS = 1600;  // radix-8 rounds per wave (a multiple of every UNROLL)
constexpr int CUS = 256;

struct Stamp {
    unsigned long long cycles, ticks;
};

// UNROLL: how many radix-8 rounds one loop iteration holds as straight-line code (1: 6 statements, 2.7 KB -- every wave of the CU runs
// the same few cache lines; 8: 48 statements, ~21 KB -- the footprint of a pass kernel's batch loop, and waves drift apart in it).
// Same work per launch either way: does the statement slow down when the instruction stream is long?
// PRIO: 0 = every wave at the default priority; 1 = s_setprio(hardware wave slot & 3): the waves that share a SIMD get DIFFERENT static
// priorities (does breaking the symmetry of the issue arbitration remove the even-occupancy penalty?)
template <int UNROLL, int PRIO = 0>
__global__ void __launch_bounds__(256, 8) k_round8(uint32_t *out, Stamp *st, int iters, uint32_t seed) {
#if defined(__HIP_DEVICE_COMPILE__)
    using namespace ntt;
    const uint64_t P = 0xFFFFFFFF00000001ull;
    uint64_t a[8], t[7];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = ((uint64_t) (threadIdx.x * 2654435761u + seed * (i + 3)) << 21 | (i * 1315423911u)) % P;
#pragma unroll
    for (int i = 0; i < 7; i++) t[i] = ((((uint64_t) (seed * 40503u + i * 97u) << 29) | (i * 2246822519u + seed)) + threadIdx.x) % P;
    if (PRIO) {
        uint32_t hw_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        switch (hw_id & 3u) {  // WAVE_ID[1:0]: the slot of this wave on its SIMD
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    // waves of a workgroup start at different places of the long body (they would drift apart anyway in a real kernel)
    for (int it = (UNROLL > 1 ? (int) (threadIdx.x >> 6) : 0); it < iters / UNROLL; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
#pragma unroll
        for (int s = 0; s < 3; s++) {
            const int h = 1 << s;
#pragma unroll
            for (int q = 0; q < 4; q += 2) {
                const int j0 = ((q >> s) << (s + 1)) | (q & (h - 1));
                const int j1 = (((q + 1) >> s) << (s + 1)) | ((q + 1) & (h - 1));
                const uint64_t t0 = t[(4 >> s) - 1 + (j0 >> (s + 1))], t1 = t[(4 >> s) - 1 + (j1 >> (s + 1))];
                gl_fwd2_v_w(a[j0], a[j0 + h], t0, a[j1], a[j1 + h], t1);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    if ((threadIdx.x & 63) == 0) st[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = Stamp{c1 - c0, r1 - r0};
    uint64_t x = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) x ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t) (x ^ (x >> 32));
#endif
}

int main() {
    uint32_t *d_out;
    Stamp *d_st;
    const size_t max_threads = (size_t) CUS * 8 * 256;
    hipMalloc(&d_out, max_threads * sizeof(uint32_t));
    hipMalloc(&d_st, max_threads / 64 * sizeof(Stamp));
    std::vector<Stamp> h(max_threads / 64);
    struct V { const char *name; void (*k)(uint32_t *, Stamp *, int, uint32_t); int unroll; };
    const V variants[] = {{"loop body = 1 round (2.7 KB)", k_round8<1>, 1}, {"loop body = 4 rounds (~11 KB)", k_round8<4>, 4}, {"loop body = 8 rounds (~21 KB)", k_round8<8>, 8},
                          {"loop body = 1 round, s_setprio(wave slot & 3)", k_round8<1, 1>, 1}};
    for (const V &v : variants) {
        auto kern = v.k;
        hipFuncAttributes fa;
        hipFuncGetAttributes(&fa, (const void *) kern);
        printf("# gl_fwd2_v_w as register-resident radix-8 rounds (12 butterflies each, %d per wave), %s; kernel: %d VGPRs\n", ITERS, v.name, fa.numRegs);
        printf("# waves/SIMD  cycles per butterfly per SIMD (median wave)  cycles per VALU instruction (22)  clock GHz  cycles per butterfly per SIMD (99th-percentile wave)\n");
        for (int w = 1; w <= 8; w++) {
            const int blocks = CUS * w;
            const size_t lds = (160 * 1024 / w) - 512;
            if (hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds) != hipSuccess) return 1;
            for (int k = 0; k < 2; k++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d_out, d_st, ITERS, 12345u + k);
            hipDeviceSynchronize();
            std::vector<double> cyc, ghz, cmax;
            for (int r = 0; r < 5; r++) {
                hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d_out, d_st, ITERS, 777u + r);
                hipDeviceSynchronize();
                const size_t nw = (size_t) blocks * 4;
                hipMemcpy(h.data(), d_st, nw * sizeof(Stamp), hipMemcpyDeviceToHost);
                std::vector<unsigned long long> c(nw), t(nw);
                for (size_t i = 0; i < nw; i++) c[i] = h[i].cycles, t[i] = h[i].ticks;
                std::nth_element(c.begin(), c.begin() + nw / 2, c.end());
                std::nth_element(t.begin(), t.begin() + nw / 2, t.end());
                cyc.push_back((double) c[nw / 2]);
                ghz.push_back((double) c[nw / 2] / ((double) t[nw / 2] * 10.0));
                // the 99th percentile of the waves' durations: with unequal priorities the waves of a SIMD finish one after the other,
                // and the SIMD's throughput is set by the LAST one (every wave starts at the same time: one generation)
                std::nth_element(c.begin(), c.begin() + (nw * 99) / 100, c.end());
                cmax.push_back((double) c[(nw * 99) / 100]);
            }
            if (hipGetLastError() != hipSuccess) return 2;
            std::sort(cyc.begin(), cyc.end());
            std::sort(ghz.begin(), ghz.end());
            std::sort(cmax.begin(), cmax.end());
            // wave w of a workgroup skips its first w iterations of the long body: median wave (1.5 skipped) ~ the nominal count
            const double rounds = (v.unroll > 1) ? (double) (ITERS / v.unroll - 1.5) * v.unroll : (double) ITERS;
            const double per_bf = cyc[2] / (w * 12.0 * rounds);
            printf("%d  %.2f  %.3f  %.3f  %.2f\n", w, per_bf, per_bf / 22.0, ghz[2], cmax[2] / (w * 12.0 * rounds));
            fflush(stdout);
        }
    }
    // ---- steady-state throughput: MANY generations of workgroups (as a pass kernel runs: a finished wave is replaced at once), W resident
    // per SIMD; cycles per butterfly per SIMD = launch duration x clock / butterflies per SIMD.  The closed-batch figures above time ONE
    // generation, whose slower waves finish alone: this is the number a long-running kernel sees.
    {
        auto kern = k_round8<1>;
        const int gens = 12, iters = 200;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        Stamp *d_st2;
        uint32_t *d_out2;
        hipMalloc(&d_st2, (size_t) CUS * 8 * gens * 4 * sizeof(Stamp));
        hipMalloc(&d_out2, (size_t) CUS * 8 * gens * 256 * sizeof(uint32_t));
        std::vector<Stamp> h2((size_t) CUS * 8 * gens * 4);
        printf("# steady state: %d generations of workgroups, %d rounds per wave, W resident per SIMD (LDS-forced)\n", gens, iters);
        printf("# waves/SIMD  cycles per butterfly per SIMD (launch duration x clock / work)  cycles per VALU instruction (22)  clock GHz\n");
        for (int w = 1; w <= 8; w++) {
            const int blocks = CUS * w * gens;
            const size_t lds = (160 * 1024 / w) - 512;
            if (hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds) != hipSuccess) return 1;
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d_out2, d_st2, iters, 99u);
            hipDeviceSynchronize();
            std::vector<double> per;
            double ghz = 0;
            for (int r = 0; r < 5; r++) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d_out2, d_st2, iters, 777u + r);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                const size_t nw = (size_t) blocks * 4;
                hipMemcpy(h2.data(), d_st2, nw * sizeof(Stamp), hipMemcpyDeviceToHost);
                std::vector<double> g(nw);
                for (size_t i = 0; i < nw; i++) g[i] = (double) h2[i].cycles / ((double) h2[i].ticks * 10.0);
                std::nth_element(g.begin(), g.begin() + nw / 2, g.end());
                ghz = g[nw / 2];
                const double bf_per_simd = (double) blocks * 4 * 12.0 * iters / (CUS * 4.0);
                per.push_back(ms * 1e-3 * ghz * 1e9 / bf_per_simd);
            }
            if (hipGetLastError() != hipSuccess) return 2;
            std::sort(per.begin(), per.end());
            printf("%d  %.2f  %.3f  %.3f\n", w, per[2], per[2] / 22.0, ghz);
            fflush(stdout);
        }
    }
    return 0;
}
