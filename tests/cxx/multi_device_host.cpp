// multi_device_host.cpp -- a C++ host that shards one [B][N] batch over every visible device through the C-ABI
// (include/ntt_hip.h), the multi-device leg of the drop-in boundary ("single process, N devices, one stream
// each").  What the reference does below its host -- scatter the data, broadcast the ONE twiddle
// table to every tile, gather (src/aie2.py:83-115) -- is here: contiguous row slabs per device, ntt_plan_clone() per
// device (device-to-device copy of the tables, no host table), one stream per device, no data-path exchange.
// Every shard is verified the reference's way (src/test.cpp:203-247): CPU network, word-by-word compare, PASS / FAIL.
// TEST program: links the oracle as the checker.  On a one-GPU box it runs with one device and must still pass;
// NTT_MD_REPLICAS=k additionally clones the plan k times onto device 0 (exercises the clone + per-stream path there).
// Built and run by tests/test_gpu_round3.py::test_cxx_multi_device_host.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ntt_hip.h"
#include "ntt_oracle.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define CHECK_NTT(x) do { int r_ = (x); if (r_ != 0) { std::printf("%s: %s\n", #x, ntt_error_string(r_)); return 1; } } while (0)

static uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

int main(int argc, char **argv) {
    const int logn = argc > 1 ? std::atoi(argv[1]) : 16;
    const size_t B = argc > 2 ? (size_t) std::atoll(argv[2]) : 37;  // ragged on purpose: shards differ in size
    const uint64_t p = 0xFFFFFFFF00000001ull, g = 7;
    const size_t N = (size_t) 1 << logn;
    int ndev = ntt_device_count();
    if (ndev < 1) { std::printf("no device\n"); return 1; }
    const char *rep = std::getenv("NTT_MD_REPLICAS");
    const int replicas = rep ? std::atoi(rep) : 0;
    // shard s lives on device dev[s]: every visible device once, plus `replicas` extra shards on device 0
    std::vector<int> dev;
    for (int d = 0; d < ndev; d++) dev.push_back(d);
    for (int r = 0; r < replicas; r++) dev.push_back(0);
    const size_t S = dev.size();
    std::printf("devices: %d, shards: %zu, N = 2^%d, batch %zu\n", ndev, S, logn, B);

    // First contact with a multi-GPU node must be readable: ntt_plan_clone copies the tables with
    // hipMemcpyPeer, which the runtime serves over xGMI when the two devices can access each other and SILENTLY stages through
    // host memory when they cannot.  Say which one every device pair is -- loudly, before anything runs -- and count them.
    int direct = 0, staged = 0;
    for (size_t s = 1; s < S; s++) {
        if (dev[s] == dev[0]) continue;  // same-device replica: an ordinary device-to-device copy
        int can = 0;
        CHECK_HIP(hipDeviceCanAccessPeer(&can, dev[s], dev[0]));
        if (can) {
            ++direct;
        } else {
            ++staged;
            std::printf("  WARNING: device %d cannot access device %d directly: the table copy of this clone is STAGED THROUGH THE HOST "
                        "(hipMemcpyPeer's fallback), not sent over xGMI\n", dev[s], dev[0]);
        }
    }
    std::printf("peer table copies: %d direct (xGMI), %d staged through the host, %zu on the source device\n", direct, staged,
                S - 1 - (size_t) direct - (size_t) staged);
    if (staged && std::getenv("NTT_MD_REQUIRE_PEER")) { std::printf("  FAIL. (NTT_MD_REQUIRE_PEER: a staged copy is an error)\n"); return 1; }

    // ONE plan with a device-generated table (no host table exists anywhere), cloned onto every other shard's device
    std::vector<ntt_plan_t> plan(S, nullptr);
    CHECK_NTT(ntt_plan_create(&plan[0], logn, p, 8, dev[0]));
    CHECK_NTT(ntt_plan_generate_twiddles(plan[0], 0, g));
    for (size_t s = 1; s < S; s++) CHECK_NTT(ntt_plan_clone(plan[0], dev[s], &plan[s]));

    // host input [B][N], a[b][i] = splitmix64(b*N + i) mod p
    std::vector<uint64_t> in(B * N), out(B * N, 0), back(B * N, 0);
    for (size_t i = 0; i < B * N; i++) in[i] = splitmix64(i) % p;

    // contiguous row slabs: the first B % S shards get one extra row
    std::vector<size_t> lo(S + 1, 0);
    for (size_t s = 0; s < S; s++) lo[s + 1] = lo[s] + B / S + (s < B % S ? 1 : 0);
    std::vector<uint64_t *> d_in(S, nullptr), d_out(S, nullptr);
    std::vector<hipStream_t> st(S);
    for (size_t s = 0; s < S; s++) {
        const size_t rows = lo[s + 1] - lo[s];
        CHECK_HIP(hipSetDevice(dev[s]));
        CHECK_HIP(hipStreamCreate(&st[s]));
        if (rows == 0) continue;
        CHECK_HIP(hipMalloc(&d_in[s], rows * N * 8));
        CHECK_HIP(hipMalloc(&d_out[s], rows * N * 8));
        CHECK_HIP(hipMemcpyAsync(d_in[s], in.data() + lo[s] * N, rows * N * 8, hipMemcpyHostToDevice, st[s]));  // scatter
    }
    // launch every shard on its own stream (asynchronous: all devices work at once), then the inverse in place
    for (size_t s = 0; s < S; s++) {
        const size_t rows = lo[s + 1] - lo[s];
        CHECK_NTT(ntt_forward(plan[s], d_in[s], d_out[s], rows, NTT_LAYOUT_NATURAL, st[s]));
        if (rows == 0) continue;
        CHECK_HIP(hipSetDevice(dev[s]));
        CHECK_HIP(hipMemcpyAsync(out.data() + lo[s] * N, d_out[s], rows * N * 8, hipMemcpyDeviceToHost, st[s]));  // gather
        CHECK_NTT(ntt_inverse(plan[s], d_out[s], d_out[s], rows, NTT_LAYOUT_NATURAL, 1, st[s]));
        CHECK_HIP(hipMemcpyAsync(back.data() + lo[s] * N, d_out[s], rows * N * 8, hipMemcpyDeviceToHost, st[s]));
    }
    for (size_t s = 0; s < S; s++) {
        CHECK_HIP(hipSetDevice(dev[s]));
        CHECK_HIP(hipStreamSynchronize(st[s]));
    }

    // verification, the reference's way: CPU network on the same input with the same table rule, word-by-word compare
    std::vector<uint64_t> roots(N), T0(N);
    oracle_make_roots_u64((uint64_t) N, roots.data(), p, g);
    CHECK_NTT(ntt_plan_get_twiddles(plan[S - 1], 0, T0.data()));  // the LAST clone's table, read back
    long errors = 0;
    for (size_t i = 1; i < N; i++) errors += T0[i] != roots[i];
    std::vector<uint64_t> ref(in);
    oracle_ntt_batch_u64(ref.data(), (uint64_t) N, B, roots.data(), p, 8);
    for (size_t s = 0; s < S; s++) {
        long e = 0;
        for (size_t i = lo[s] * N; i < lo[s + 1] * N; i++) e += (ref[i] != out[i]) + (back[i] != in[i]);
        std::printf("  shard %zu on device %d: rows [%zu, %zu)  %s\n", s, dev[s], lo[s], lo[s + 1], e ? "MISMATCH" : "ok");
        errors += e;
    }
    for (size_t s = 0; s < S; s++) {
        (void) hipSetDevice(dev[s]);
        if (d_in[s]) (void) hipFree(d_in[s]);
        if (d_out[s]) (void) hipFree(d_out[s]);
        (void) hipStreamDestroy(st[s]);
        ntt_plan_destroy(plan[s]);
    }
    if (!errors) { std::printf("  PASS!\n"); return 0; }
    std::printf("  mismatches: %ld\n  FAIL.\n", errors);
    return 1;
}
