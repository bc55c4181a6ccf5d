// kernels_m64_inv.hip -- pass kernels for FieldM64 (any odd p < 2^64, Montgomery R = 2^64), inverse network (see pass.h).
#define NTT_FIELD ntt::FieldM64
#define NTT_INV true
#define NTT_LAUNCH_FN launch_m64_inv
#include "pass_kernel.inc"
