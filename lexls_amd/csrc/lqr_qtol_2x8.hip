// levels of 8 rows, n + 1 <= 32 columns (n read from the arguments), x only, tolerance contract
#include "lqr_qtol_impl.h"
LEXLS_QTOL_INSTANCE(launch_qtol_2x8, 2, 8, 0, 0)
