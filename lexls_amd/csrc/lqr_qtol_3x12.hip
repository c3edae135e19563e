// levels of 12 rows, 33 <= n + 1 <= 48 columns other than the IK shape's 41 (n read from the arguments), x only, tolerance contract
#include "lqr_qtol_impl.h"
LEXLS_QTOL_INSTANCE(launch_qtol_3x12, 3, 12, 0, 0)
