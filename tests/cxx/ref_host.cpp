// ref_host.cpp -- a C++ host that drives libntt_hip.so the way the reference host drives its
// XRT kernel, and verifies like it: CPU network (here the oracle,
// this is a TEST program), block-order permutation, word-by-word compare, PASS/FAIL exit code.
// Built and run by tests/test_gpu_parity.py::test_cxx_host_through_c_abi.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "ntt_hip.h"
#include "ntt_oracle.h"

int main() {
    constexpr int logn = 11;
    constexpr uint64_t p = 3329, g = 3;
    const size_t N = (size_t) 1 << logn;

    ntt_plan_t plan = nullptr;
    int rc = ntt_plan_create(&plan, logn, p, 4, 0);
    if (rc) { std::printf("plan: %s\n", ntt_error_string(rc)); return 1; }
    std::vector<uint32_t> root(N), in(N), out(N, 0);
    ntt_make_roots(plan, g, root.data());
    for (size_t i = 0; i < N; i++) in[i] = (uint32_t) i;
    rc = ntt_plan_set_twiddles(plan, root.data());
    if (rc) { std::printf("twiddles: %s\n", ntt_error_string(rc)); return 1; }

    uint32_t *d_in = nullptr, *d_out = nullptr;
    if (hipMalloc(&d_in, N * 4) != hipSuccess || hipMalloc(&d_out, N * 4) != hipSuccess) return 1;
    (void) hipMemcpy(d_in, in.data(), N * 4, hipMemcpyHostToDevice);
    (void) hipMemset(d_out, 0, N * 4);

    std::printf("Running Kernel.\n");
    for (int i = 0; i < 10; i++) {
        auto start = std::chrono::high_resolution_clock::now();
        rc = ntt_forward(plan, d_in, d_out, 1, NTT_LAYOUT_AIE_BLOCK16, nullptr);
        (void) hipStreamSynchronize(nullptr);
        auto stop = std::chrono::high_resolution_clock::now();
        if (rc) { std::printf("kernel did not complete: %s\n", ntt_error_string(rc)); return 1; }
        (void) hipMemcpy(out.data(), d_out, N * 4, hipMemcpyDeviceToHost);
        std::printf("%lld\n", (long long) std::chrono::duration_cast<std::chrono::microseconds>(stop - start).count());
    }

    // CPU reference + block order, as the reference's verification does
    std::vector<uint32_t> a_ref(in), roots_cpu(N), answers(N);
    oracle_make_roots_u32((uint32_t) N, roots_cpu.data(), (uint32_t) p, (uint32_t) g);
    oracle_ntt_u32(a_ref.data(), (uint32_t) N, roots_cpu.data(), (uint32_t) p, logn - 1);
    oracle_block16_u32(answers.data(), a_ref.data(), (uint32_t) N);
    int errors = 0;
    for (size_t i = 0; i < N; i++) errors += (root[i] != roots_cpu[i]) + (answers[i] != out[i]);
    std::printf("  logN: %d\n  p: %llu\n", logn, (unsigned long long) p);
    ntt_plan_destroy(plan);
    if (!errors) { std::printf("  PASS!\n"); return 0; }
    std::printf("  mismatches: %d\n  FAIL.\n", errors);
    return 1;
}
