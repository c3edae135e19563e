#include "lqr_quad_impl.h"
LEXLS_QUAD_INSTANCE(launch_quad_3x12_x, 3, 12, false, 0)
