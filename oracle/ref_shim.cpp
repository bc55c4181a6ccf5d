// ref_shim.cpp -- C-ABI doorway onto the LITERAL reference code.
//
// TEST INFRASTRUCTURE ONLY.  This file holds none of the reference's text: the
// build recipe (oracle/build_ref.sh) extracts line ranges from the read-only
// reference sources where they lie and hands them in through -D...=<tmpfile>;
// the temporary files live outside the repository and are deleted after the
// compile.  The result, oracle/_ref/libntt_ref.so, is git-ignored.
//
// Why line ranges: src/test.cpp as a file needs XRT, boost and mlir-aie's
// test_utils.h, none of which exist in this image, so the file is unbuildable
// here; the functions on the hot path (test.cpp:15-60) and the scalar
// arithmetic twins (aie_core.cc:11-39) are self-contained and compile as they
// stand.  No stand-in header, library or generated code is written for them.
//
//   REF_TEST_15_60   src/test.cpp:15-60    modPow, make_roots, ntt
//   REF_TEST_69_71   src/test.cpp:69-71    block_num, ans_order
//   REF_TEST_212_219 src/test.cpp:212-219  answers[] block permutation
//   REF_CORE_11_39   src/aie_core.cc:11-39 modadd, modsub, barrett_2k
#include <array>
#include <cstdint>
#include <cstring>
#include <vector>

#include REF_TEST_15_60
#include REF_CORE_11_39

extern "C" {

int32_t ref_modPow(int32_t x, int32_t n, int32_t mod) { return modPow(x, n, mod); }

// test.cpp:137-139: std::vector<int32_t> root(N); root[0] = 1; make_roots(...)
void ref_make_roots(int32_t n, int32_t *roots_out, int32_t p, int32_t g) {
    std::vector<int32_t> root(n);
    root[0] = 1;
    make_roots(n, root, p, g);
    std::memcpy(roots_out, root.data(), sizeof(int32_t) * (size_t) n);
}

// test.cpp:207: ntt(a_ref, IN_VOLUME, root, p, test_stage)
void ref_ntt(int32_t *a_inout, int32_t n, const int32_t *roots, int32_t p, int32_t stage) {
    std::vector<int32_t> a(a_inout, a_inout + n);
    std::vector<int32_t> r(roots, roots + n);
    ntt(a, n, r, p, stage);
    std::memcpy(a_inout, a.data(), sizeof(int32_t) * (size_t) n);
}

// test.cpp:212-219 with the declarations of :69-71 in scope.
void ref_block_order(int32_t *answers_out, const int32_t *a_ref_in, int32_t IN_VOLUME) {
    std::vector<int32_t> a_ref(a_ref_in, a_ref_in + IN_VOLUME);
#include REF_TEST_69_71
#include REF_TEST_212_219
    std::memcpy(answers_out, answers.data(), sizeof(int32_t) * (size_t) IN_VOLUME);
}

int32_t ref_modadd(int32_t a, int32_t b, int32_t q) { return modadd(a, b, q); }
int32_t ref_modsub(int32_t a, int32_t b, int32_t q) { return modsub(a, b, q); }
int32_t ref_barrett_2k(int32_t a, int32_t b, int32_t q, int32_t w, int32_t u) {
    return barrett_2k(a, b, q, w, u);
}

}  // extern "C"
