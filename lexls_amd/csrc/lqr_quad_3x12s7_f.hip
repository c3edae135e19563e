#include "lqr_quad_impl.h"
LEXLS_QUAD_INSTANCE(launch_quad_3x12s7_f, 3, 12, true, 7)
