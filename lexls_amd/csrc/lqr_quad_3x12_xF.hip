// lqr_quad<3,12> with fixed variables (FIX), x only, layout offset 0
#include "lqr_quad_impl.h"
LEXLS_QUAD_INSTANCE_FIX(launch_quad_3x12_xF, 3, 12, false, 0)
