// levels of 8 rows, 33 <= n + 1 <= 48 columns (n read from the arguments), x only, tolerance contract
#include "lqr_qtol_impl.h"
LEXLS_QTOL_INSTANCE(launch_qtol_3x8, 3, 8, 0, 0)
