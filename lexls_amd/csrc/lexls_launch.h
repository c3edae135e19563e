// Host-side launchers of the liblexls_hip kernels (one per translation unit that defines kernels).
#pragma once
#include "lexls_kernels.h"

namespace lexls
{
    /// maximum dynamic LDS one workgroup may ask for on gfx950 (160 KiB per CU)
    constexpr size_t kMaxLdsBytes = 160 * 1024;

    /// odd leading dimension >= rows for the LDS image (conflict-free column-per-lane access)
    inline uint32_t odd_ld(uint32_t rows) { return rows | 1u; }

    // lqr_generic.hip — any shape, one workgroup per problem
    hipError_t launch_lqr_generic(LseArgs a, uint32_t max_rows, bool write_factor, bool do_solve, hipStream_t s, const char **variant);
    hipError_t launch_solve_generic(const LseArgs &a, hipStream_t s, bool reciprocal_diagonal = false);
    hipError_t launch_residual(const LseArgs &a, hipStream_t s);
    hipError_t launch_sensitivity(const LseArgs &a, const int32_t *d_obj_index, int32_t obj_all, double tolW, double tolC, hipStream_t s, bool scan_up = false,
                                  uint32_t sweep_level_dim_hint = 0); // hint = largest level dimension of the batch (enables the single-sweep kernel)
    bool sensitivity_sweep_serves(const LseArgs &a, uint32_t sweep_level_dim_hint); // launch_sensitivity takes the one-wavefront-per-problem sweep for these arguments
    hipError_t launch_leastnorm(const LseArgs &a, hipStream_t s);
    hipError_t launch_leastnorm2(const LseArgs &a, hipStream_t s);
    hipError_t launch_leastnorm3(const LseArgs &a, hipStream_t s);
    hipError_t launch_gather_rows(const LseArgs &a, const double *d_cdata, uint64_t per_problem, const uint32_t *d_row_src, const uint32_t *d_row_ld,
                                  double *d_dst, hipStream_t s);

    // lqr_small.hip — one wavefront per problem, problem in VGPRs (n+1 <= 64, rows <= 64, level dims <= 16)
    bool wave_kernel_supports(const LseArgs &a, uint32_t max_rows, uint32_t max_level_dim, bool has_fixed);
    bool deep_kernel_supports(const LseArgs &a, uint32_t max_level_dim, bool write_factor, bool has_fixed);
    bool wave_dispatch_is_register_resident(const LseArgs &a, uint32_t max_level_dim, bool has_fixed, int left_looking);
    /// tolerance: x-only solves of the shapes lqr_mfma_impl.h / lqr_qtol_impl.h serve may take those kernels (pivots / ranks exact, x within 1e-10
    /// instead of bit-identical to the oracle).  0 = bit-exact kernels only; 1 = automatic (lqr_mfma where it serves, else lqr_qtol); 6 = lqr_qtol
    /// only; 7 / 8 = lqr_mfma with two / one problem per wavefront, else lqr_qtol
    hipError_t launch_lqr_wave(const LseArgs &a, uint32_t max_level_dim, bool write_factor, bool has_fixed, int left_looking, hipStream_t s,
                               const char **variant, int tolerance = 0);

    /// The resident active-set iterations of a lock-step LexLSI batch as one persistent launch (lsi_fused_impl.h): l-QR (the register-resident wave
    /// kernel's body, rows gathered by reference) -> removal sweep -> iteration, per instance until it stops or `count` iterations are done.
    /// resident_args: the driver's ResidentArgs (lexls_lsi_device.h).  hipErrorNotSupported: the shape has no persistent instantiation (the caller
    /// enqueues the three kernels per stage instead); the conditions are those under which launch_lqr_wave(a, ..., factor kept, left_looking < 0)
    /// takes the same register-resident instantiation and launch_sensitivity the sweep
    hipError_t launch_lsi_fused(const LseArgs &a, uint32_t max_level_dim, bool has_fixed, const int32_t *d_obj_index, double tolW, double tolC, bool scan_up,
                                const void *resident_args, size_t resident_args_bytes, int count, hipStream_t s, const char **variant);

    // lqr_large.hip — problems too large for one CU's LDS: one launch per stage, the whole chip per problem
    bool generic_fits_lds(const LseArgs &a, uint32_t max_rows);
    bool large_kernel_supports(const LseArgs &a, uint32_t max_level_dim, bool has_fixed);
    size_t large_state_bytes(uint32_t batch);
    hipError_t launch_lqr_large(const LseArgs &a, const uint32_t *h_level_max, uint32_t h_rows_max, void *d_state, double *d_norms, hipStream_t s);
    size_t large_fast_workspace_bytes(uint32_t batch, uint32_t n, uint32_t cap, uint32_t maxdim);
    hipError_t launch_lqr_large_fast(const LseArgs &a, const uint32_t *h_level_max, uint32_t h_rows_max, void *d_workspace, hipStream_t s);
} // namespace lexls
