// levels of 12 rows, any n with n + 1 <= 48, x only, tolerance contract, one problem per wavefront (four wavefronts per SIMD)
#include "lqr_mfma_impl.h"
LEXLS_MFMA_INSTANCE(launch_mfma_64x12, 64, 12, 0)
