// guard.h -- the exception wall of the C-ABI (include/ntt_hip.h: "nothing throws or aborts").
//
// Every extern "C" entry point of ntt_api.hip is a function-try-block:
//
//     int ntt_forward(...) NTT_GUARD {
//         ...
//     } NTT_GUARD_END
//
// so a std::bad_alloc from a host-side std::vector (the table staging buffers are N words: 2 GiB each at
// logn 28) or any other C++ exception becomes an error CODE instead of crossing the C boundary, where it
// would terminate a C / Go / Python caller.  The reference's error contract is the same shape: a failed
// run prints and returns 1, it never aborts the host (src/test.cpp:162-166).
//
// Host-only header (no HIP types): tests/cxx/guard_test.cpp compiles it with g++ on the CPU.
#pragma once
#include <new>

// the enum values of include/ntt_hip.h (ntt_api.hip static_asserts that they agree)
#define NTT_E_NOMEM_GUARD (-9)
#define NTT_E_INTERNAL_GUARD (-10)

#define NTT_GUARD try
#define NTT_GUARD_END                                      \
    catch (const std::bad_alloc &) {                       \
        return NTT_E_NOMEM_GUARD;                          \
    }                                                      \
    catch (...) {                                          \
        return NTT_E_INTERNAL_GUARD;                       \
    }
