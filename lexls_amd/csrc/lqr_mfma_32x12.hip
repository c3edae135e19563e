// levels of 12 rows, any n with n + 1 <= 48, x only, tolerance contract, two problems per wavefront
#include "lqr_mfma_impl.h"
LEXLS_MFMA_INSTANCE(launch_mfma_32x12, 32, 12, 0)
