// lqr_quad<3,12> with fixed variables (FIX), factor kept, layout offset 0
#include "lqr_quad_impl.h"
LEXLS_QUAD_INSTANCE_FIX(launch_quad_3x12_fF, 3, 12, true, 0)
