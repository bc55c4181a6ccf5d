// kernels_m64_product.hip -- the fused middle pass of the negacyclic product (pass.h: run_product_pass) for the general odd 64-bit
// modulus (FieldM64): the same radix-8 schedule as kernels_gl_product.hip, unit sizes 2^7 .. 2^12.
#include "product_kernel.inc"

namespace ntt {

bool have_m64_product_mid(int log_m) { return log_m >= 7 && log_m <= 12; }

bool m64_product_mid_fits(int log_m, int n, uint32_t batch, uint32_t target_wgs) {
    switch (log_m) {
        case 7: return product_fits<ProductCfg<7, FieldM64>>(n, batch, target_wgs);
        case 8: return product_fits<ProductCfg<8, FieldM64>>(n, batch, target_wgs);
        case 9: return product_fits<ProductCfg<9, FieldM64>>(n, batch, target_wgs);
        case 10: return product_fits<ProductCfg<10, FieldM64>>(n, batch, target_wgs);
        case 11: return product_fits<ProductCfg<11, FieldM64>>(n, batch, target_wgs);
        case 12: return product_fits<ProductCfg<12, FieldM64>>(n, batch, target_wgs);
        default: return false;
    }
}

hipError_t launch_m64_product_mid(int log_m, const ErasedArgs &a, hipStream_t s) {
    switch (log_m) {
        case 7: return launch_product<ProductCfg<7, FieldM64>>(a, s);
        case 8: return launch_product<ProductCfg<8, FieldM64>>(a, s);
        case 9: return launch_product<ProductCfg<9, FieldM64>>(a, s);
        case 10: return launch_product<ProductCfg<10, FieldM64>>(a, s);
        case 11: return launch_product<ProductCfg<11, FieldM64>>(a, s);
        case 12: return launch_product<ProductCfg<12, FieldM64>>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ntt
