// guard_test.cpp -- the exception wall of the C-ABI (ntt_aie_amd/csrc/guard.h) on the CPU: a function-try-block built
// from NTT_GUARD / NTT_GUARD_END turns std::bad_alloc into NTT_E_NOMEM and anything else into NTT_E_INTERNAL, and leaves
// ordinary return values alone.  Compiled with g++ by tests/test_abi.py (no HIP needed: guard.h is host-only).
#include <stdint.h>
#include <stdio.h>

#include <stdexcept>
#include <vector>

#include "../../include/ntt_hip.h"
#include "../../ntt_aie_amd/csrc/guard.h"

static_assert(NTT_E_NOMEM == NTT_E_NOMEM_GUARD && NTT_E_INTERNAL == NTT_E_INTERNAL_GUARD, "codes agree with the header");

extern "C" {
int plain(int v) NTT_GUARD { return v; } NTT_GUARD_END
int throws_bad_alloc(void) NTT_GUARD { throw std::bad_alloc(); } NTT_GUARD_END
// what ntt_plan_set_twiddles does at a size the host cannot serve: a std::vector of "N words"
int real_vector_too_large(void) NTT_GUARD {
    std::vector<uint64_t> v((size_t) 1 << 59);  // 4 EiB: fails in operator new, not in the OOM killer
    return (int) v.size();
} NTT_GUARD_END
int throws_length_error(void) NTT_GUARD {
    std::vector<uint64_t> v;
    v.resize(v.max_size() + 1);  // std::length_error: not a bad_alloc
    return 0;
} NTT_GUARD_END
int throws_int(void) NTT_GUARD { throw 7; } NTT_GUARD_END
int64_t wide(void) NTT_GUARD { throw std::runtime_error("x"); } NTT_GUARD_END
}

int main() {
    printf("plain=%d bad_alloc=%d vector=%d length=%d int=%d wide=%lld\n", plain(42), throws_bad_alloc(), real_vector_too_large(),
           throws_length_error(), throws_int(), (long long) wide());
    return 0;
}
