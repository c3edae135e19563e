// lqr_quad<1,12>: n + 1 <= 16 columns in 1 slot(s), factor kept (the small IK families: fewer live slots per pivot step)
#include "lqr_quad_impl.h"
LEXLS_QUAD_INSTANCE(launch_quad_1x12_f, 1, 12, true, 0)
