// kernels_gl_product.hip -- the fused middle pass of the Goldilocks negacyclic product (pass.h: run_product_pass):
// last inverse-network pass of both operands + pointwise product + first forward-network pass in one workgroup-resident
// sweep over each 2^LOG_M-word unit (no reference counterpart: the reference has no product).
#include "product_kernel.inc"

namespace ntt {

bool have_gl_product_mid(int log_m) { return log_m >= 7 && log_m <= 12; }

bool gl_product_mid_fits(int log_m, int n, uint32_t batch, uint32_t target_wgs) {
    switch (log_m) {
        case 7: return product_fits<ProductCfg<7>>(n, batch, target_wgs);
        case 8: return product_fits<ProductCfg<8>>(n, batch, target_wgs);
        case 9: return product_fits<ProductCfg<9>>(n, batch, target_wgs);
        case 10: return product_fits<ProductCfg<10>>(n, batch, target_wgs);
        case 11: return product_fits<ProductCfg<11>>(n, batch, target_wgs);
        case 12: return product_fits<ProductCfg<12>>(n, batch, target_wgs);
        default: return false;
    }
}

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
