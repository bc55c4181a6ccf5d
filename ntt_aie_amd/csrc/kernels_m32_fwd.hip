// kernels_m32_fwd.hip -- pass kernels for FieldM32, forward network (see pass.h).
#define NTT_FIELD ntt::FieldM32
#define NTT_INV false
#define NTT_LAUNCH_FN launch_m32_fwd
#include "pass_kernel.inc"
