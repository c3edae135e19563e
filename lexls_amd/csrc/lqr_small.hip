// Dispatch of the one-wavefront-per-problem kernels (lqr_small_impl.h); each instantiation lives in its own
// translation unit lqr_small_<shape>.hip so that they compile in parallel.
#include "lexls_kernels.h"
#include "lexls_launch.h"

namespace lexls
{
    hipError_t launch_wave_41x12e_x(const LseArgs &a, hipStream_t s);
    hipError_t launch_wave_41x12e_f(const LseArgs &a, hipStream_t s);
    hipError_t launch_wave_41x12_x(const LseArgs &a, hipStream_t s);
    hipError_t launch_wave_41x12_f(const LseArgs &a, hipStream_t s);
    hipError_t launch_wave_64x16_x(const LseArgs &a, hipStream_t s);
    hipError_t launch_wave_64x16_f(const LseArgs &a, hipStream_t s);

    hipError_t launch_lwave_41x12e_x(const LseArgs &a, hipStream_t s);
    hipError_t launch_lwave_41x12e_f(const LseArgs &a, hipStream_t s);
    hipError_t launch_lwave_41x12_x(const LseArgs &a, hipStream_t s);
    hipError_t launch_lwave_41x12_f(const LseArgs &a, hipStream_t s);

    bool wave_kernel_supports(const LseArgs &a, uint32_t max_rows, uint32_t max_level_dim, bool has_fixed)
    {
        (void)has_fixed; // fixed variables are handled in-kernel
        return a.nVar + 1 <= 64 && max_rows <= 64 && max_level_dim <= 16 && a.nObj <= 16;
    }

    hipError_t launch_lqr_wave(const LseArgs &a, uint32_t max_level_dim, bool write_factor, bool has_fixed, bool left_looking, hipStream_t s,
                               const char **variant)
    {
        const uint32_t nc = a.nVar + 1;
        // left-looking form (lqr_lwave_impl.h): one level block live per wave, 4 waves/SIMD; no fixed variables
        if (left_looking && !has_fixed && max_level_dim <= 12 && nc <= 41 && a.nObj <= 8)
        {
            if (nc == 41)
            {
                *variant = "lqr_lwave<41,12,exact>";
                return write_factor ? launch_lwave_41x12e_f(a, s) : launch_lwave_41x12e_x(a, s);
            }
            *variant = "lqr_lwave<41,12>";
            return write_factor ? launch_lwave_41x12_f(a, s) : launch_lwave_41x12_x(a, s);
        }
        if (max_level_dim <= 12 && nc == 41)
        {
            *variant = "lqr_wave<41,12,exact>";
            return write_factor ? launch_wave_41x12e_f(a, s) : launch_wave_41x12e_x(a, s);
        }
        if (max_level_dim <= 12 && nc <= 41)
        {
            *variant = "lqr_wave<41,12>";
            return write_factor ? launch_wave_41x12_f(a, s) : launch_wave_41x12_x(a, s);
        }
        *variant = "lqr_wave<64,16>";
        return write_factor ? launch_wave_64x16_f(a, s) : launch_wave_64x16_x(a, s);
    }
} // namespace lexls
