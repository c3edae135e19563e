// the IK shape (n = 40, levels of 12 rows), x only, tolerance contract: the bench kernel
#include "lqr_qtol_impl.h"
LEXLS_QTOL_INSTANCE(launch_qtol_3x12s7, 3, 12, 7, 40)
