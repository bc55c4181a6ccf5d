// kernels_m32_inv.hip -- pass kernels for FieldM32, inverse network (see pass.h).
#define NTT_FIELD ntt::FieldM32
#define NTT_INV true
#define NTT_LAUNCH_FN launch_m32_inv
#include "pass_kernel.inc"
