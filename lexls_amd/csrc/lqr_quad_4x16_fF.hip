// lqr_quad<4,16> with fixed variables (FIX) and the factor kept, layout offset 0
#include "lqr_quad_impl.h"
LEXLS_QUAD_INSTANCE_FIX(launch_quad_4x16_fF, 4, 16, true, 0)
