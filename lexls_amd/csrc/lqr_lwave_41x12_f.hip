#include "lqr_lwave_impl.h"
LEXLS_LWAVE_INSTANCE(launch_lwave_41x12_f, 41, 12, false, true)
