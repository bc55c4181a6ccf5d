// kernels_gl_inv.hip -- pass kernels for FieldGL, inverse network (see pass.h).
#define NTT_FIELD ntt::FieldGL
#define NTT_INV true
#define NTT_LAUNCH_FN launch_gl_inv
#include "pass_kernel.inc"
