// levels of 12 rows, n + 1 <= 32 columns, x only, tolerance contract
#include "lqr_qtol_impl.h"
LEXLS_QTOL_INSTANCE(launch_qtol_2x12, 2, 12, 0, 0)
