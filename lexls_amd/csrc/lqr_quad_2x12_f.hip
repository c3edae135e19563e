// lqr_quad<2,12>: n + 1 <= 32 columns in 2 slot(s), factor kept (the small IK families: fewer live slots per pivot step)
#include "lqr_quad_impl.h"
LEXLS_QUAD_INSTANCE(launch_quad_2x12_f, 2, 12, true, 0)
