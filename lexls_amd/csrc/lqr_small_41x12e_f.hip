#include "lqr_small_impl.h"
LEXLS_WAVE_INSTANCE(launch_wave_41x12e_f, 41, 12, true, true)
