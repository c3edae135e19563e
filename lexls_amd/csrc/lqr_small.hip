// Dispatch of the one-wavefront-per-problem kernels (lqr_small_impl.h); each instantiation lives in its own
// translation unit lqr_small_<shape>.hip so that they compile in parallel.
#include "lexls_kernels.h"
#include "lexls_launch.h"

namespace lexls
{
    hipError_t launch_wave_41x12e_x(const LseArgs &a, hipStream_t s);
    hipError_t launch_wave_41x12e_f(const LseArgs &a, hipStream_t s);
#define LEXLS_DECLARE_LSI_FUSED(NAME)                                                                                                                            \
    hipError_t NAME(const LseArgs &a, uint32_t sweep_level_dim, const int32_t *d_obj_index, double tolW, double tolC, bool scan_up, const void *resident_args, \
                    size_t resident_args_bytes, int count, hipStream_t s);
    LEXLS_DECLARE_LSI_FUSED(launch_lsi_fused_41x12e)
    LEXLS_DECLARE_LSI_FUSED(launch_lsi_fused_41x12)
    LEXLS_DECLARE_LSI_FUSED(launch_lsi_fused_64x16)
#undef LEXLS_DECLARE_LSI_FUSED
    hipError_t launch_wave_41x12_x(const LseArgs &a, hipStream_t s);
    hipError_t launch_wave_41x12_f(const LseArgs &a, hipStream_t s);
    hipError_t launch_wave_64x16_x(const LseArgs &a, hipStream_t s);
    hipError_t launch_wave_64x16_f(const LseArgs &a, hipStream_t s);
    hipError_t launch_wave_41x12_fR(const LseArgs &a, hipStream_t s);
    hipError_t launch_wave_64x16_fR(const LseArgs &a, hipStream_t s);

    hipError_t launch_lwave_41x12e_x(const LseArgs &a, hipStream_t s);
    hipError_t launch_lwave_41x12e_f(const LseArgs &a, hipStream_t s);
    hipError_t launch_lwave_41x12_x(const LseArgs &a, hipStream_t s);
    hipError_t launch_lwave_41x12_f(const LseArgs &a, hipStream_t s);

    hipError_t launch_quad_3x12_x(const LseArgs &a, hipStream_t s);
    hipError_t launch_quad_4x16_x(const LseArgs &a, hipStream_t s);
    hipError_t launch_quad_4x16_f(const LseArgs &a, hipStream_t s);
    hipError_t launch_quad_4x16_fF(const LseArgs &a, hipStream_t s);
    hipError_t launch_quad_3x12s7_x(const LseArgs &a, hipStream_t s);
    hipError_t launch_quad_3x12_f(const LseArgs &a, hipStream_t s);
    hipError_t launch_quad_3x12s7_f(const LseArgs &a, hipStream_t s);
    hipError_t launch_quad_2x12_x(const LseArgs &a, hipStream_t s);
    hipError_t launch_quad_1x12_x(const LseArgs &a, hipStream_t s);
    hipError_t launch_quad_2x12_f(const LseArgs &a, hipStream_t s);
    hipError_t launch_quad_1x12_f(const LseArgs &a, hipStream_t s);
    hipError_t launch_quad_3x12_xF(const LseArgs &a, hipStream_t s);
    hipError_t launch_quad_3x12s7_xF(const LseArgs &a, hipStream_t s);
    hipError_t launch_quad_4x16_xF(const LseArgs &a, hipStream_t s);
    hipError_t launch_quad_3x12_fF(const LseArgs &a, hipStream_t s);
    hipError_t launch_quad_3x12s7_fF(const LseArgs &a, hipStream_t s);
    size_t quad_lds_bytes(uint32_t slots, uint32_t md, uint32_t nVar, uint32_t nObj);
    hipError_t launch_qtol_3x12s7(const LseArgs &a, hipStream_t s);
    size_t launch_qtol_3x12s7_lds(uint32_t nVar, uint32_t nObj);
    hipError_t launch_qtol_3x12(const LseArgs &a, hipStream_t s);
    size_t launch_qtol_3x12_lds(uint32_t nVar, uint32_t nObj);
    hipError_t launch_qtol_2x12(const LseArgs &a, hipStream_t s);
    size_t launch_qtol_2x12_lds(uint32_t nVar, uint32_t nObj);
    hipError_t launch_qtol_3x8(const LseArgs &a, hipStream_t s);
    size_t launch_qtol_3x8_lds(uint32_t nVar, uint32_t nObj);
    hipError_t launch_qtol_2x8(const LseArgs &a, hipStream_t s);
    size_t launch_qtol_2x8_lds(uint32_t nVar, uint32_t nObj);
    hipError_t launch_mfma_32x12n40(const LseArgs &a, hipStream_t s);
    size_t launch_mfma_32x12n40_lds(uint32_t nVar, uint32_t nObj);
    hipError_t launch_mfma_32x12(const LseArgs &a, hipStream_t s);
    size_t launch_mfma_32x12_lds(uint32_t nVar, uint32_t nObj);
    hipError_t launch_mfma_16x12n40(const LseArgs &a, hipStream_t s);
    hipError_t launch_mfma_64x12(const LseArgs &a, hipStream_t s);
    size_t launch_mfma_64x12_lds(uint32_t nVar, uint32_t nObj);

    /// which instantiation of the matrix-core tolerance-contract kernel (lqr_mfma_impl.h) serves these arguments (0: none): x-only solves of
    /// batches in which every level of every problem has exactly 12 rows, no fixed variables, no regularization, n + 1 <= 48 —
    /// 1: two problems per wavefront, n = 40 (the IK shape of BASELINE configs[2]/[3]); 2: two problems per wavefront, other n;
    /// 3: one problem per wavefront (asked for by policy 8)
    static int mfma_choice(const LseArgs &a, bool write_factor, bool has_fixed, bool one_per_wave)
    {
        if (write_factor || has_fixed || a.reg_type != 0 || a.uniform_dim != 12 || a.nObj > 8 || (a.cap & 1u) != 0 || (reinterpret_cast<uintptr_t>(a.in) & 15u) != 0 || a.g_cdata) return 0;
        if (a.nVar < 1 || a.nVar + 1 > 48) return 0;
        if (one_per_wave) return 4 * launch_mfma_64x12_lds(a.nVar, a.nObj) <= kMaxLdsBytes ? 3 : 0;
        if (2 * launch_mfma_32x12_lds(a.nVar, a.nObj) > kMaxLdsBytes) return 0;
        return a.nVar == 40 ? 1 : 2;
    }

    /// which instantiation of the tolerance-contract four-per-wavefront kernel (lqr_qtol_impl.h) serves these arguments (0: none): x-only
    /// solves of batches in which every level of every problem has exactly 12 (or exactly 8) rows, no fixed variables, no regularization, n + 1 <= 48 —
    /// 1: n = 40, the IK shape of BASELINE configs[2]/[3] (n a compile-time constant, columns right-aligned in the slots); 2: other n with
    /// 33 .. 48 columns; 3: up to 32 columns
    static int qtol_choice(const LseArgs &a, bool write_factor, bool has_fixed)
    {
        if (write_factor || has_fixed || a.reg_type != 0 || (a.uniform_dim != 12 && a.uniform_dim != 8) || a.nObj > 8 || (a.cap & 1u) != 0 || (reinterpret_cast<uintptr_t>(a.in) & 15u) != 0 || a.g_cdata) return 0;
        if (a.uniform_dim == 8) // levels of eight rows (round 4): 4: 33 .. 48 columns, 5: up to 32
        {
            if (a.nVar + 1 <= 32) return (a.nVar >= 2 && launch_qtol_2x8_lds(a.nVar, a.nObj) <= kMaxLdsBytes) ? 5 : 0;
            if (a.nVar + 1 <= 48) return launch_qtol_3x8_lds(a.nVar, a.nObj) <= kMaxLdsBytes ? 4 : 0;
            return 0;
        }
        if (a.nVar == 40) return launch_qtol_3x12s7_lds(a.nVar, a.nObj) <= kMaxLdsBytes ? 1 : 0;
        if (a.nVar + 1 <= 32) return (a.nVar >= 2 && launch_qtol_2x12_lds(a.nVar, a.nObj) <= kMaxLdsBytes) ? 3 : 0;
        if (a.nVar + 1 <= 48) return launch_qtol_3x12_lds(a.nVar, a.nObj) <= kMaxLdsBytes ? 2 : 0;
        return 0;
    }

    bool wave_kernel_supports(const LseArgs &a, uint32_t max_rows, uint32_t max_level_dim, bool has_fixed)
    {
        (void)has_fixed; // fixed variables are handled in-kernel
        if (a.reg_type == 7) return false; // the experimental type's by-products need the level lists: generic kernel (lexls_regularize.h)
        return a.nVar + 1 <= 64 && max_rows <= 64 && max_level_dim <= 16 && a.nObj <= 16;
    }

    /// waves of the register-resident kernel the device holds at once (2 per SIMD): up to that many problems run in ONE round of it
    static uint32_t resident_wave_capacity()
    {
        static uint32_t cap_of[64] = {0}; // per device (a process may drive several)
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256u * 4u * 2u;
        if (!cap_of[dev])
        {
            int cus = 0;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
            cap_of[dev] = (uint32_t)cus * 4u * 2u;
        }
        return cap_of[dev];
    }

    /// which four-per-wavefront instantiation (0: none) automatic dispatch / the policies take for these arguments
    static int quad_choice(const LseArgs &a, uint32_t max_level_dim, bool write_factor, int left_looking, bool has_fixed)
    {
        if (a.reg_type != 0) return 0;
        // Measured on MI355X (scripts/crossover.py, n = 40, 5 x 12, us per batch, register-resident / four-per-wavefront): x only — 512: 58 / 53,
        // 1024: 62 / 54, 2048: 78 / 58, 4096: 146 / 63: the four-per-wavefront kernel at every batch size; factor kept — 1024: 71 / 98,
        // 2048: 89 / 103, 3072: 144 / 109, 4096: 166 / 115: the register-resident kernel while the batch fits one round of it.
        const bool lwave_pays = left_looking > 0 || (left_looking == 0 && a.batch > resident_wave_capacity());
        if (left_looking < 0 || left_looking == 1) return 0; // register-resident / left-looking wave kernel asked for
        // Factor kept, automatic dispatch (scripts/dispatch_scan.py, register-resident / four-per-wavefront, us per batch): one slot (n = 12,
        // 3 x 4) 33 / 28 at 256, 42 / 29 at 2048: four-per-wavefront at every batch size; two slots (n = 20, 30) within 10 % up to 1024,
        // 62 / 56 and 66 / 56 at 2048: from 2048 on; 42..48 columns, where the register-resident alternative is the 64-column
        // instantiation (n = 47, 4 x 12): 120 / 96 at 256, 299 / 109 at 2048: at every batch size
        const uint32_t nc = a.nVar + 1;
        const bool full   = a.batch >= resident_wave_capacity(); // the register-resident kernel is at two wavefronts per SIMD
        const bool forced = left_looking == 2;
        size_t lds = (max_level_dim <= 12) ? quad_lds_bytes(3, 12, a.nVar, a.nObj) : 0;
        if (lds && lds <= kMaxLdsBytes)
        {
            const bool take = forced || !write_factor || lwave_pays || (!has_fixed && nc <= 16) || (!has_fixed && nc <= 32 && full) || nc > 41;
            if (!take) return 0;
            if (!has_fixed && nc <= 16) return 5; // one slot
            if (!has_fixed && nc <= 32) return 4; // two slots
            return a.nVar == 40 ? 2 : 1; // 2: the IK shape, columns right-aligned in the slots (see SIG in lqr_quad_impl.h)
        }
        // n + 1 <= 64, level dims <= 16: x only at every batch size; with the factor kept when forced (deep hierarchies, kernel policy 4) or when
        // the batch needs more than one round of the register-resident kernel (n = 55, [16,14,16,12], factor kept, us per batch, register-
        // resident / four-per-wavefront: 1024: 168 / 183, 2048: 332 / 206, 4096: 640 / 404, 8192: 1155 / 799: from 2048 on)
        lds = (max_level_dim <= 16 && (!write_factor || forced || lwave_pays || full)) ? quad_lds_bytes(4, 16, a.nVar, a.nObj) : 0;
        if (lds && lds <= kMaxLdsBytes) return 3;
        return 0;
    }

    /// true when launch_lqr_wave (same arguments, factor kept) takes the register-resident lqr_wave_kernel — the one whose load can gather
    /// the rows by reference (LseArgs::g_cdata); the left-looking and four-per-wavefront kernels read an assembled problem
    bool wave_dispatch_is_register_resident(const LseArgs &a, uint32_t max_level_dim, bool has_fixed, int left_looking)
    {
        if (a.reg_type != 0) return true;
        if (quad_choice(a, max_level_dim, true, left_looking, has_fixed) != 0) return false;
        const bool lwave_pays = left_looking > 0 || (left_looking == 0 && a.batch > resident_wave_capacity());
        const uint32_t nc     = a.nVar + 1;
        return !(lwave_pays && !has_fixed && max_level_dim <= 12 && nc <= 41 && a.nObj <= 8);
    }

    /// Deep hierarchies — more than 64 rows in all, which the register-resident kernel's LDS image of the rows below a level cannot hold —
    /// are served by the left-looking kernels (a level's rows come from HBM when the level starts; LDS holds only the finished pivot rows,
    /// at most nVar of them): true when launch_lqr_wave with left_looking = 2 ends in lqr_quad or lqr_lwave for these arguments
    bool deep_kernel_supports(const LseArgs &a, uint32_t max_level_dim, bool write_factor, bool has_fixed)
    {
        if (a.reg_type != 0 || a.nVar + 1 > 64) return false;
        if (quad_choice(a, max_level_dim, write_factor, 2, has_fixed) != 0) return true;
        return !has_fixed && max_level_dim <= 12 && a.nVar + 1 <= 41 && a.nObj <= 8;
    }

    hipError_t launch_lqr_wave(const LseArgs &a, uint32_t max_level_dim, bool write_factor, bool has_fixed, int left_looking, hipStream_t s,
                               const char **variant, int tolerance)
    {
        const uint32_t nc = a.nVar + 1;
        // tolerance: 0 bit-exact kernels only; 1 automatic (lqr_qtol where it serves — the faster of the two on MI355X: 41 us against 57 us per
        // 4096 IK problems — else the matrix-core kernel); 6 lqr_qtol only; 7 / 8 / 9 lqr_mfma with two / one / four problems per wavefront
        // (9: the IK shape only), else lqr_qtol
        auto try_mfma = [&](bool one_per_wave) -> int { return mfma_choice(a, write_factor, has_fixed, one_per_wave); };
        if (tolerance == 9 && try_mfma(false) == 1)
        {
            *variant = "lqr_mfma<16,12,n40>";
            return launch_mfma_16x12n40(a, s);
        }
        if (tolerance == 7 || tolerance == 8)
            switch (try_mfma(tolerance == 8))
            {
            case 1: *variant = "lqr_mfma<32,12,n40>"; return launch_mfma_32x12n40(a, s);
            case 2: *variant = "lqr_mfma<32,12>"; return launch_mfma_32x12(a, s);
            case 3: *variant = "lqr_mfma<64,12>"; return launch_mfma_64x12(a, s);
            default: break;
            }
        if (tolerance != 0)
            switch (qtol_choice(a, write_factor, has_fixed))
            {
            case 1: *variant = "lqr_qtol<3,12,shift 7>"; return launch_qtol_3x12s7(a, s);
            case 2: *variant = "lqr_qtol<3,12>"; return launch_qtol_3x12(a, s);
            case 3: *variant = "lqr_qtol<2,12>"; return launch_qtol_2x12(a, s);
            case 4: *variant = "lqr_qtol<3,8>"; return launch_qtol_3x8(a, s);
            case 5: *variant = "lqr_qtol<2,8>"; return launch_qtol_2x8(a, s);
            default: break;
            }
        if (tolerance == 1) // shapes lqr_qtol's four slices per wavefront do not hold
            switch (try_mfma(false))
            {
            case 1: *variant = "lqr_mfma<32,12,n40>"; return launch_mfma_32x12n40(a, s);
            case 2: *variant = "lqr_mfma<32,12>"; return launch_mfma_32x12(a, s);
            default: break;
            }
        if (a.reg_type != 0) // the regularization family: the register-resident kernel's REG instantiations (factor always kept)
        {
            if (max_level_dim <= 12 && nc <= 41)
            {
                *variant = "lqr_wave<41,12,regularized>";
                return launch_wave_41x12_fR(a, s);
            }
            *variant = "lqr_wave<64,16,regularized>";
            return launch_wave_64x16_fR(a, s);
        }
        // left-looking form (lqr_lwave_impl.h): one level block live per wave, 4 waves/SIMD; no fixed variables.  Measured on MI355X
        // (scripts/latency_scan.py, n = 40, 5 x 12): while the batch fits one round of the register-resident kernel (<= 2048 problems)
        // that kernel has the shorter latency (factor kept: 66-72 us vs 97-101 us; x only: equal); beyond that the left-looking kernel
        // still runs in one round (4096: 115 us vs 158 us).  left_looking: 0 = decide by batch size, > 0 = always, < 0 = never
        const bool lwave_pays = left_looking > 0 || (left_looking == 0 && a.batch > resident_wave_capacity());
        // four problems per wavefront (lqr_quad_impl.h): one wave per SIMD serves 4 x 4 x CUs problems per round; fixed variables in the FIX
        // instantiations.  left_looking == 2 forces it (parity tests)
        switch (quad_choice(a, max_level_dim, write_factor, left_looking, has_fixed))
        {
        case 5:
            *variant = write_factor ? "lqr_quad<1,12,factor>" : "lqr_quad<1,12>";
            return write_factor ? launch_quad_1x12_f(a, s) : launch_quad_1x12_x(a, s);
        case 4:
            *variant = write_factor ? "lqr_quad<2,12,factor>" : "lqr_quad<2,12>";
            return write_factor ? launch_quad_2x12_f(a, s) : launch_quad_2x12_x(a, s);
        case 2:
            *variant = has_fixed ? (write_factor ? "lqr_quad<3,12,shift 7,factor,fixed>" : "lqr_quad<3,12,shift 7,fixed>")
                                 : (write_factor ? "lqr_quad<3,12,shift 7,factor>" : "lqr_quad<3,12,shift 7>");
            if (has_fixed) return write_factor ? launch_quad_3x12s7_fF(a, s) : launch_quad_3x12s7_xF(a, s);
            return write_factor ? launch_quad_3x12s7_f(a, s) : launch_quad_3x12s7_x(a, s);
        case 1:
            *variant = has_fixed ? (write_factor ? "lqr_quad<3,12,factor,fixed>" : "lqr_quad<3,12,fixed>") : (write_factor ? "lqr_quad<3,12,factor>" : "lqr_quad<3,12>");
            if (has_fixed) return write_factor ? launch_quad_3x12_fF(a, s) : launch_quad_3x12_xF(a, s);
            return write_factor ? launch_quad_3x12_f(a, s) : launch_quad_3x12_x(a, s);
        case 3:
            if (write_factor)
            {
                *variant = has_fixed ? "lqr_quad<4,16,factor,fixed>" : "lqr_quad<4,16,factor>";
                return has_fixed ? launch_quad_4x16_fF(a, s) : launch_quad_4x16_f(a, s);
            }
            *variant = has_fixed ? "lqr_quad<4,16,fixed>" : "lqr_quad<4,16>";
            return has_fixed ? launch_quad_4x16_xF(a, s) : launch_quad_4x16_x(a, s);
        default: break;
        }
        if (lwave_pays && !has_fixed && max_level_dim <= 12 && nc <= 41 && a.nObj <= 8)
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

    hipError_t launch_lsi_fused(const LseArgs &a, uint32_t max_level_dim, bool has_fixed, const int32_t *d_obj_index, double tolW, double tolC, bool scan_up,
                                const void *resident_args, size_t resident_args_bytes, int count, hipStream_t s, const char **variant)
    {
        const uint32_t nc = a.nVar + 1;
        if (a.reg_type != 0 || !a.g_cdata || !wave_dispatch_is_register_resident(a, max_level_dim, has_fixed, -1) || !sensitivity_sweep_serves(a, max_level_dim))
            return hipErrorNotSupported;
        // (the register-resident instantiation launch_lqr_wave takes for these arguments)
        if (max_level_dim <= 12 && nc == 41)
        {
            *variant = "lsi_fused<lqr_wave<41,12,exact>>";
            return launch_lsi_fused_41x12e(a, max_level_dim, d_obj_index, tolW, tolC, scan_up, resident_args, resident_args_bytes, count, s);
        }
        if (max_level_dim <= 12 && nc <= 41)
        {
            *variant = "lsi_fused<lqr_wave<41,12>>";
            return launch_lsi_fused_41x12(a, max_level_dim, d_obj_index, tolW, tolC, scan_up, resident_args, resident_args_bytes, count, s);
        }
        *variant = "lsi_fused<lqr_wave<64,16>>";
        return launch_lsi_fused_64x16(a, max_level_dim, d_obj_index, tolW, tolC, scan_up, resident_args, resident_args_bytes, count, s);
    }
} // namespace lexls
