#include "lsi_fused_impl.h"
LEXLS_LSI_FUSED_INSTANCE(launch_lsi_fused_64x16, 64, 16, false)
