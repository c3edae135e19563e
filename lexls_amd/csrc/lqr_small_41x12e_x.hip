#include "lqr_small_impl.h"
LEXLS_WAVE_INSTANCE(launch_wave_41x12e_x, 41, 12, true, false)
