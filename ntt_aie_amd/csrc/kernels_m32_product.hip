// kernels_m32_product.hip -- the same fused middle pass for 4-byte words (any odd p < 2^32): radix-16 rounds, unit sizes
// 2^6 .. 2^13 (used; 2^5 instantiated for the host-model test); for N <= 2^13 the launch is the whole negacyclic product.
#include "product_kernel.inc"

namespace ntt {

// (2^5 has a kernel too, but its register loads and stores move 8 bytes per polynomial at a time: 3.9 ms per GiB of operands
// against 1.4 ms for the three separate launches, whose small units are staged through LDS -- tools/polymul_small.py)
bool have_m32_product_mid(int log_m) { return log_m >= 6 && log_m <= 13; }

bool m32_product_mid_fits(int log_m, int n, uint32_t batch, uint32_t target_wgs) {
    switch (log_m) {
        case 6: return product_fits<ProductCfgM32<6>>(n, batch, target_wgs);
        case 7: return product_fits<ProductCfgM32<7>>(n, batch, target_wgs);
        case 8: return product_fits<ProductCfgM32<8>>(n, batch, target_wgs);
        case 9: return product_fits<ProductCfgM32<9>>(n, batch, target_wgs);
        case 10: return product_fits<ProductCfgM32<10>>(n, batch, target_wgs);
        case 11: return product_fits<ProductCfgM32<11>>(n, batch, target_wgs);
        case 12: return product_fits<ProductCfgM32<12>>(n, batch, target_wgs);
        case 13: return product_fits<ProductCfgM32<13>>(n, batch, target_wgs);
        default: return false;
    }
}

hipError_t launch_m32_product_mid(int log_m, const ErasedArgs &a, hipStream_t s) {
    switch (log_m) {
        case 5: return launch_product<ProductCfgM32<5>>(a, s);
        case 6: return launch_product<ProductCfgM32<6>>(a, s);
        case 7: return launch_product<ProductCfgM32<7>>(a, s);
        case 8: return launch_product<ProductCfgM32<8>>(a, s);
        case 9: return launch_product<ProductCfgM32<9>>(a, s);
        case 10: return launch_product<ProductCfgM32<10>>(a, s);
        case 11: return launch_product<ProductCfgM32<11>>(a, s);
        case 12: return launch_product<ProductCfgM32<12>>(a, s);
        case 13: return launch_product<ProductCfgM32<13>>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ntt
