// lqr_quad<3,12> with fixed variables (FIX), factor kept, layout offset 7
#include "lqr_quad_impl.h"
LEXLS_QUAD_INSTANCE_FIX(launch_quad_3x12s7_fF, 3, 12, true, 7)
