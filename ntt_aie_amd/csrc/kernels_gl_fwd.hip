// kernels_gl_fwd.hip -- pass kernels for FieldGL, forward network (see pass.h).
#define NTT_FIELD ntt::FieldGL
#define NTT_INV false
#define NTT_LAUNCH_FN launch_gl_fwd
#include "pass_kernel.inc"
