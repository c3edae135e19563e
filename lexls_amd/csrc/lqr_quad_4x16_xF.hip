// lqr_quad<4,16> with fixed variables (FIX), x only, layout offset 0
#include "lqr_quad_impl.h"
LEXLS_QUAD_INSTANCE_FIX(launch_quad_4x16_xF, 4, 16, false, 0)
