#include "lqr_small_impl.h"
LEXLS_WAVE_INSTANCE(launch_wave_64x16_x, 64, 16, false, false)
