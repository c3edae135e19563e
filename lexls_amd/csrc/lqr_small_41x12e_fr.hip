// lqr_wave<41,12,exact,factor> for RAGGED batches (the equality problems of a lock-step LexLSI stage: level dims follow the working sets and
// are rarely the capacity): only the run-time-guarded form of every level, half the code (68 KB instead of 112 KB against a 64 KB
// instruction cache) — measured 84.3 us vs 88.4 us per 1024 ragged problems.
#define LEXLS_WAVE_NO_FULL
#include "lqr_small_impl.h"
LEXLS_WAVE_INSTANCE(launch_wave_41x12e_fr, 41, 12, true, true)
