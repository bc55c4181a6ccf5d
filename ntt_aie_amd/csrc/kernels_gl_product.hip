// kernels_gl_product.hip -- the fused middle pass of the Goldilocks negacyclic product (pass.h: run_product_pass):
// last inverse-network pass of both operands + pointwise product + first forward-network pass in one workgroup-resident
// sweep over each 2^LOG_M-word unit (no reference counterpart: the reference has no product).
#include "product_kernel.inc"

namespace ntt {

bool have_gl_product_mid(int log_m) { return log_m >= 7 && log_m <= 12; }

hipError_t launch_gl_product_mid(int log_m, const ErasedArgs &a, hipStream_t s) {
    switch (log_m) {
        case 7: return launch_product<ProductCfg<7>>(a, s);
        case 8: return launch_product<ProductCfg<8>>(a, s);
        case 9: return launch_product<ProductCfg<9>>(a, s);
        case 10: return launch_product<ProductCfg<10>>(a, s);
        case 11: return launch_product<ProductCfg<11>>(a, s);
        case 12: return launch_product<ProductCfg<12>>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ntt
