// lqr_wave<64,16> with the regularization family (REG): factor kept, nVar <= 63, level dims <= 16
#include "lqr_small_impl.h"
LEXLS_WAVE_INSTANCE_REG(launch_wave_64x16_fR, 64, 16)
