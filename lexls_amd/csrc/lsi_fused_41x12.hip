#include "lsi_fused_impl.h"
LEXLS_LSI_FUSED_INSTANCE(launch_lsi_fused_41x12, 41, 12, false)
