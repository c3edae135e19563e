// the IK shape (n = 40, levels of 12 rows), x only, tolerance contract, four problems per wavefront (one wavefront per SIMD)
#include "lqr_mfma_impl.h"
LEXLS_MFMA_INSTANCE(launch_mfma_16x12n40, 16, 12, 40)
