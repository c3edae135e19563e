#include "lqr_small_impl.h"
LEXLS_WAVE_INSTANCE(launch_wave_41x12_x, 41, 12, false, false)
