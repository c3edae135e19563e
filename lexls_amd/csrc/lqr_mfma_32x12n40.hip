// the IK shape (n = 40, levels of 12 rows), x only, tolerance contract, two problems per wavefront: the bench kernel
#include "lqr_mfma_impl.h"
LEXLS_MFMA_INSTANCE(launch_mfma_32x12n40, 32, 12, 40)
