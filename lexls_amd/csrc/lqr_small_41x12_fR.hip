// lqr_wave<41,12> with the regularization family (REG): factor kept, nVar <= 40, level dims <= 12
#include "lqr_small_impl.h"
LEXLS_WAVE_INSTANCE_REG(launch_wave_41x12_fR, 41, 12)
