// the IK shape itself (n = 40): 41 columns right-aligned in the 48 lanes-slots of a row
#include "lqr_quad_impl.h"
LEXLS_QUAD_INSTANCE(launch_quad_3x12s7_x, 3, 12, false, 7)
