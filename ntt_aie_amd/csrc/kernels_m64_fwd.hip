// kernels_m64_fwd.hip -- pass kernels for FieldM64 (any odd p < 2^64, Montgomery R = 2^64), forward network (see pass.h).
#define NTT_FIELD ntt::FieldM64
#define NTT_INV false
#define NTT_LAUNCH_FN launch_m64_fwd
#include "pass_kernel.inc"
