// Device helpers shared by the one-wavefront-per-problem kernels (lqr_small_impl.h, lqr_lwave_impl.h).
#pragma once
#include "lexls_kernels.h"
#include "lexls_launch.h"

#include <cfloat>
#include <type_traits>

namespace lexls
{
    namespace
    {
        __device__ __forceinline__ double dfma(double a, double b, double c) { return __builtin_fma(a, b, c); }

        __device__ __forceinline__ double rdlane(double v, int lane)
        {
            const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
            const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
            return __hiloint2double(hi, lo);
        }

        /// max of two doubles as ONE v_max_f64 (the builtin adds canonicalising v_max x,x around it; keys are never NaN)
        __device__ __forceinline__ double vmax(double x, double y)
        {
            double r;
            asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
            return r;
        }

        template <int CTRL>
        __device__ __forceinline__ double dpp_max(double v)
        {
            // source-only DPP moves (no tied "old" operand -> no register copies); every lane of the 16-lane row is written
            const int lo2 = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
            const int hi2 = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
            return vmax(v, __hiloint2double(hi2, lo2));
        }

        template <int CTRL>
        __device__ __forceinline__ int dpp_maxi(int v)
        {
            const int o = __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); // folds into v_max_i32_dpp
            return o > v ? o : v;
        }
        /// signed maximum over the 64 lanes, wave-uniform: butterfly inside the 16-lane DPP rows, then v_permlane16_swap / v_permlane32_swap (both
        /// operands the same value: whichever rows the instruction exchanges, the two results hold the two partners of every lane)
        __device__ __forceinline__ int wave_maxi(int v)
        {
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            v = dpp_maxi<0xB1>(v);  // quad_perm [1,0,3,2]
            v = dpp_maxi<0x4E>(v);  // quad_perm [2,3,0,1]
            v = dpp_maxi<0x141>(v); // row_half_mirror
            v = dpp_maxi<0x140>(v); // row_mirror
            const u2 r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
            v          = (int)r.x > (int)r.y ? (int)r.x : (int)r.y;
            const u2 q = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
            v          = (int)q.x > (int)q.y ? (int)q.x : (int)q.y;
            return __builtin_amdgcn_readfirstlane(v);
        }

        /// maximum over the 64 lanes, returned as a wave-uniform value: butterfly inside each 16-lane DPP row, then the four
        /// row results are combined through SGPRs (two more DPP steps — row_bcast:15 / row_bcast:31 — need fewer instructions
        /// but lengthen the dependent chain: measured slower)
        __device__ __forceinline__ double wave_max(double v)
        {
            v = dpp_max<0xB1>(v);  // quad_perm [1,0,3,2]
            v = dpp_max<0x4E>(v);  // quad_perm [2,3,0,1]
            v = dpp_max<0x141>(v); // row_half_mirror
            v = dpp_max<0x140>(v); // row_mirror
            const double r0 = rdlane(v, 0), r1 = rdlane(v, 16), r2 = rdlane(v, 32), r3 = rdlane(v, 48);
            return vmax(vmax(r0, r1), vmax(r2, r3)); // three v_max_f64 (fmax() would canonicalise every operand first)
        }

        /// T[idx] for a wave-uniform idx: a scalar branch tree instead of dynamic register indexing
        template <int NC>
        __device__ __forceinline__ double select_reg(const double (&T)[NC], int idx)
        {
            double v = 0.0;
            switch (idx)
            {
#define LEXLS_CASE(J)            \
    case J:                      \
        if (J < NC) v = T[J < NC ? J : 0]; \
        break;
                LEXLS_CASE(0) LEXLS_CASE(1) LEXLS_CASE(2) LEXLS_CASE(3) LEXLS_CASE(4) LEXLS_CASE(5) LEXLS_CASE(6) LEXLS_CASE(7)
                LEXLS_CASE(8) LEXLS_CASE(9) LEXLS_CASE(10) LEXLS_CASE(11) LEXLS_CASE(12) LEXLS_CASE(13) LEXLS_CASE(14) LEXLS_CASE(15)
                LEXLS_CASE(16) LEXLS_CASE(17) LEXLS_CASE(18) LEXLS_CASE(19) LEXLS_CASE(20) LEXLS_CASE(21) LEXLS_CASE(22) LEXLS_CASE(23)
                LEXLS_CASE(24) LEXLS_CASE(25) LEXLS_CASE(26) LEXLS_CASE(27) LEXLS_CASE(28) LEXLS_CASE(29) LEXLS_CASE(30) LEXLS_CASE(31)
                LEXLS_CASE(32) LEXLS_CASE(33) LEXLS_CASE(34) LEXLS_CASE(35) LEXLS_CASE(36) LEXLS_CASE(37) LEXLS_CASE(38) LEXLS_CASE(39)
                LEXLS_CASE(40) LEXLS_CASE(41) LEXLS_CASE(42) LEXLS_CASE(43) LEXLS_CASE(44) LEXLS_CASE(45) LEXLS_CASE(46) LEXLS_CASE(47)
                LEXLS_CASE(48) LEXLS_CASE(49) LEXLS_CASE(50) LEXLS_CASE(51) LEXLS_CASE(52) LEXLS_CASE(53) LEXLS_CASE(54) LEXLS_CASE(55)
                LEXLS_CASE(56) LEXLS_CASE(57) LEXLS_CASE(58) LEXLS_CASE(59) LEXLS_CASE(60) LEXLS_CASE(61) LEXLS_CASE(62) LEXLS_CASE(63)
#undef LEXLS_CASE
            default: break;
            }
            return v;
        }

        template <int NC>
        __device__ __forceinline__ void store_reg(double (&T)[NC], int idx, double val, bool pred)
        {
            switch (idx)
            {
#define LEXLS_CASE(J)                                   \
    case J:                                             \
        if (J < NC && pred) T[J < NC ? J : 0] = val;    \
        break;
                LEXLS_CASE(0) LEXLS_CASE(1) LEXLS_CASE(2) LEXLS_CASE(3) LEXLS_CASE(4) LEXLS_CASE(5) LEXLS_CASE(6) LEXLS_CASE(7)
                LEXLS_CASE(8) LEXLS_CASE(9) LEXLS_CASE(10) LEXLS_CASE(11) LEXLS_CASE(12) LEXLS_CASE(13) LEXLS_CASE(14) LEXLS_CASE(15)
                LEXLS_CASE(16) LEXLS_CASE(17) LEXLS_CASE(18) LEXLS_CASE(19) LEXLS_CASE(20) LEXLS_CASE(21) LEXLS_CASE(22) LEXLS_CASE(23)
                LEXLS_CASE(24) LEXLS_CASE(25) LEXLS_CASE(26) LEXLS_CASE(27) LEXLS_CASE(28) LEXLS_CASE(29) LEXLS_CASE(30) LEXLS_CASE(31)
                LEXLS_CASE(32) LEXLS_CASE(33) LEXLS_CASE(34) LEXLS_CASE(35) LEXLS_CASE(36) LEXLS_CASE(37) LEXLS_CASE(38) LEXLS_CASE(39)
                LEXLS_CASE(40) LEXLS_CASE(41) LEXLS_CASE(42) LEXLS_CASE(43) LEXLS_CASE(44) LEXLS_CASE(45) LEXLS_CASE(46) LEXLS_CASE(47)
                LEXLS_CASE(48) LEXLS_CASE(49) LEXLS_CASE(50) LEXLS_CASE(51) LEXLS_CASE(52) LEXLS_CASE(53) LEXLS_CASE(54) LEXLS_CASE(55)
                LEXLS_CASE(56) LEXLS_CASE(57) LEXLS_CASE(58) LEXLS_CASE(59) LEXLS_CASE(60) LEXLS_CASE(61) LEXLS_CASE(62) LEXLS_CASE(63)
#undef LEXLS_CASE
            default: break;
            }
        }

        __device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

        // ---- a problem per 16-lane DPP row (lqr_quad_impl.h, the removal-search sweep of lqr_generic.hip) ----
        extern "C" __device__ double lexls_update_dpp_f64(double, double, int, int, int, bool) __asm("llvm.amdgcn.update.dpp.f64");

        /// value of lane L of this lane's 16-lane row (v_mov_b64_dpp row_newbcast:L)
        template <int L>
        __device__ __forceinline__ double gbc(double v)
        {
            return lexls_update_dpp_f64(0.0, v, 0x150 + L, 0xf, 0xf, true);
        }
        template <int L>
        __device__ __forceinline__ int gbci(int v)
        {
            return __builtin_amdgcn_update_dpp(0, v, 0x150 + L, 0xf, 0xf, true);
        }

        /// maximum over the 16 lanes of a DPP row, in every lane of the row
        __device__ __forceinline__ double row_max16(double v)
        {
            v = dpp_max<0xB1>(v);  // quad_perm [1,0,3,2]
            v = dpp_max<0x4E>(v);  // quad_perm [2,3,0,1]
            v = dpp_max<0x141>(v); // row_half_mirror
            v = dpp_max<0x140>(v); // row_mirror
            return v;
        }
        template <int CTRL>
        __device__ __forceinline__ unsigned dpp_minu(unsigned v)
        {
            const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true); // folds into v_min_u32_dpp
            return o < v ? o : v;
        }
        __device__ __forceinline__ unsigned row_min16(unsigned v)
        {
            v = dpp_minu<0xB1>(v);
            v = dpp_minu<0x4E>(v);
            v = dpp_minu<0x141>(v);
            v = dpp_minu<0x140>(v);
            return v;
        }
        /// max / min over the four rows of a value that is uniform inside each row
        __device__ __forceinline__ int rows_max(int v)
        {
            const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16), c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
            const int ab = a > b ? a : b, cd = c > d ? c : d;
            return ab > cd ? ab : cd;
        }
        __device__ __forceinline__ int rows_min(int v)
        {
            const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16), c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
            const int ab = a < b ? a : b, cd = c < d ? c : d;
            return ab < cd ? ab : cd;
        }

        /// c ? x : y on VALUES (the built-in operator on two lvalues yields an lvalue: a select of addresses, which keeps arrays in memory)
        template <class T>
        __device__ __forceinline__ T sel(bool c, T x, T y)
        {
            return c ? x : y;
        }


        /// f(integral_constant<int, I>) for I = B .. E-1: a loop whose index is a compile-time constant in the body
        template <int B, int E, class F>
        __device__ __forceinline__ void for_each_index(F &&f)
        {
            if constexpr (B < E)
            {
                f(std::integral_constant<int, B>{});
                for_each_index<B + 1, E>(f);
            }
        }

// Diagnostic build only (-DLEXLS_WAVE_STAMPS): per-phase shader-clock totals of every wave go to the (otherwise unused)
// lambda buffer; the product build contains no stamp.  Phases: 0 load, 1 transpose, 2 pivot search, 3 norms+rank test,
// 4 householder scalars (exchange, sqrt, division), 5 apply+downdate, 6 image store, 7 trsm, 8 gemm, 9 solve, 10 output.
#ifdef LEXLS_WAVE_STAMPS
#define STAMP_DECL                   \
    unsigned long long st_acc[11];   \
    for (int i_ = 0; i_ < 11; i_++) st_acc[i_] = 0; \
    unsigned long long st_t0 = clock64();
#define STAMP(i)                                   \
    {                                              \
        const unsigned long long t_ = clock64();   \
        st_acc[i] += t_ - st_t0;                   \
        st_t0 = t_;                                \
    }
#define STAMP_WRITE                                                                              \
    if (lane == 0)                                                                               \
        for (int i_ = 0; i_ < 11; i_++) a.lambda[(size_t)b * (n + cap) + i_] = (double)st_acc[i_];
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_WRITE
#endif

    } // namespace
} // namespace lexls
