// lqr_quad<4,16> with the factor kept (deep hierarchies with level dimensions 13..16, kernel policy 4), layout offset 0
#include "lqr_quad_impl.h"
LEXLS_QUAD_INSTANCE(launch_quad_4x16_f, 4, 16, true, 0)
