// Regularization family of LexLSE::factorize (lexlse.h:277-411, :1700-2251, :2592-2625) for the generic kernel: executed by the
// whole workgroup right after a level's QR, before its Gauss step.  Implemented: TIKHONOV (1), TIKHONOV_CG (2), R (3), R_NO_Z (4),
// RT_NO_Z (5), RT_NO_Z_CG (6), TIKHONOV_2 (8), TEST (9) and the variable factor; the reference's experimental TIKHONOV_1 (7,
// regularize_tikhonov_1_test :1774-1886 with its X_mu / residual_mu by-products) in the generic kernel only.
// Arithmetic order = oracle/lexlse_oracle.h (regularize_* there), so the results are bit-identical to the oracle's.
//
// Per problem the scratch holds (doubles): NS n x (n+1) [the accumulated null-space basis, lexlse.h:93; it survives the
// factorization because solveLeastNorm_3 reads it], D n x n, D0 n x n, d n, out n, 8 scalars.
#pragma once
#include "lexls_kernels.h"

namespace lexls
{
    namespace
    {
        __device__ __forceinline__ double rfma(double a, double b, double c) { return __builtin_fma(a, b, c); }

        __host__ __device__ inline size_t reg_scratch_doubles(uint32_t n) { return (size_t)n * (n + 1) + 2 * (size_t)n * n + 2 * (size_t)n + 8 + 10 * (size_t)n; }

        /// by-products of the experimental type 7 per problem (lexlse.h:96-99): X_mu nObj x n (a column per level, contiguous),
        /// X_mu_rhs nObj x n, residual_mu cap, one work vector cap
        __host__ __device__ inline size_t reg_mu_doubles(uint32_t n, uint32_t nObj, uint32_t cap) { return 2 * (size_t)nObj * n + 2 * (size_t)cap; }

        /// what regularize_tikhonov_1_test reads besides the level itself: first rows / first columns / ranks of the levels so far,
        /// the column transpositions, the Householder scalars
        struct RegLevels
        {
            const uint32_t *fr, *fc, *rk, *perm;
            const double *hh;
            uint32_t dim;
        };

// Diagnostic build only (-DLEXLS_WAVE_STAMPS): shader-clock totals of the routines' phases, accumulated by thread 0 behind the kernel's own stamps
#ifdef LEXLS_WAVE_STAMPS
#define REG_STAMP(v, i, t)                                \
    if ((v).stamp && (t) == 0)                            \
    {                                                     \
        const double now_ = (double)clock64();            \
        (v).stamp[i] += now_ - (v).stamp[15];             \
        (v).stamp[15] = now_;                             \
    }
#else
#define REG_STAMP(v, i, t)
#endif

        struct RegView
        {
            double *stamp = nullptr;
            double fk     = 0.0;   // the level's factor when the caller fetched it ahead (has_fk)
            bool has_fk   = false;
            double *W;   // the problem matrix (LDS or HBM), column-major
            size_t ld;
            uint32_t n, nf;
            double *NS;  // n x (n+1), leading dimension ldns (n in the handle's scratch; odd in an LDS window: rows of a column pair fall on different banks)
            double *D, *D0, *d, *out, *scal, *cg; // cg: 10 n doubles for the CGLS vectors
            uint32_t ldns, ldd;
            double *Dg;   // the work matrix in the handle's scratch (ld = n): what D points at when the order exceeds the LDS window
            double *Dl;   // LDS window ldl x ldl of the register-resident kernel (NULL: none)
            uint32_t ldl;
            __device__ double &ns(uint32_t i, uint32_t j) const { return NS[i + (size_t)j * ldns]; }
            __device__ double &w(uint32_t i, uint32_t j) const { return W[i + j * ld]; }
            __device__ double &dd(uint32_t i, uint32_t j) const { return D[i + (size_t)j * ldd]; }
            __device__ double &d0(uint32_t i, uint32_t j) const { return D0[i + (size_t)j * n]; }
            /// the work matrix of order N: the LDS window when it holds it (uniform per workgroup)
            __device__ RegView with_order(uint32_t N) const
            {
                RegView r = *this;
                if (Dl && N <= ldl)
                {
                    r.D   = Dl;
                    r.ldd = ldl;
                }
                else
                {
                    r.D   = Dg;
                    r.ldd = n;
                }
                return r;
            }
        };

        __device__ inline RegView reg_view(const LseArgs &a, uint32_t b, double *W, size_t ld, uint32_t nf)
        {
            RegView v;
            const uint32_t n = a.nVar;
            double *base     = a.reg_scratch + (size_t)b * reg_scratch_doubles(n);
            v.W    = W;
            v.ld   = ld;
            v.n    = n;
            v.nf   = nf;
            v.NS   = base;
            v.ldns = n;
            v.D    = base + (size_t)n * (n + 1);
            v.Dg   = v.D;
            v.ldd  = n;
            v.Dl   = nullptr;
            v.ldl  = 0;
            v.D0   = v.D + (size_t)n * n;
            v.d    = v.D0 + (size_t)n * n;
            v.out  = v.d + n;
            v.scal = v.out + n;
            v.cg   = v.scal + 8;
            return v;
        }

        __device__ __forceinline__ double reg_rdlane(double v, int lane)
        {
            const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
            const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
            return __hiloint2double(hi, lo);
        }

        /// acc + sum of x[i]^2, i = 0 .. len-1 in that order, by ONE wavefront (len <= 64): one load per lane, then the ordered fma chain on
        /// values read across lanes — every lane ends with the same sum (what thread 0's loop over the vector gave, without its load per term)
        __device__ __forceinline__ double reg_wave_sumsq(const double *x, uint32_t len, uint32_t lane, double acc)
        {
            const double xi = lane < len ? x[lane] : 0.0;
            for (uint32_t i = 0; i < len; i++)
            {
                const double t = reg_rdlane(xi, (int)i);
                acc            = rfma(t, t, acc);
            }
            return acc;
        }

        /// reg_cholesky_solve for ONE wavefront (the register-resident kernel), N <= 64: lane i owns row i of the lower triangle and d_i.
        /// Panels of 16 columns at a time in registers; a finished column's entries reach the other lanes by v_readlane, the columns of
        /// earlier panels come back from the work matrix as the lane's OWN row (no lane reads what another lane wrote until the backward
        /// substitution) — so the 3 N dependent steps (square root / division each) cost their arithmetic, not a memory round trip plus a loop
        /// of unknown trip count each, which is where the routine's time went.  Same operations on the same operands in the same order as
        /// the workgroup form below (entry (i, j) receives the products k = 0 .. j-1 in that order; substitutions column by column), so the
        /// results are the same bits.  The panel loops have run-time trip counts and a fixed body: ~70 registers, a few KB of code.
        __device__ __noinline__ void reg_cholesky_solve_wave(const RegView &v, uint32_t N, uint32_t lane)
        {
            constexpr int PB = 16;
            const bool on    = lane < N;
            double row[PB];
            for (uint32_t c0 = 0; c0 < N; c0 += PB)
            {
#pragma unroll
                for (int jj = 0; jj < PB; jj++) row[jj] = (on && c0 + jj <= lane && c0 + jj < N) ? v.dd(lane, c0 + jj) : 0.0;
                for (uint32_t k = 0; k < c0; k++) // the columns of the panels before: own entry from the matrix, the panel rows' entries from their lanes
                {
                    const double lik = (on && lane >= c0) ? v.dd(lane, k) : 0.0;
#pragma unroll
                    for (int jj = 0; jj < PB; jj++) row[jj] = rfma(-lik, reg_rdlane(lik, (int)(c0 + jj) & 63), row[jj]);
                }
#pragma unroll
                for (int kk = 0; kk < PB; kk++)
                    if (c0 + kk < N) // uniform
                    {
                        const double lkk = sqrt(reg_rdlane(row[kk], (int)(c0 + kk)));
                        const double lik = (lane == c0 + kk) ? lkk : row[kk] / lkk; // (zeros above the diagonal and beyond N)
                        row[kk]          = lik;
#pragma unroll
                        for (int jj = kk + 1; jj < PB; jj++) row[jj] = rfma(-lik, reg_rdlane(lik, (int)(c0 + jj) & 63), row[jj]);
                    }
#pragma unroll
                for (int jj = 0; jj < PB; jj++)
                    if (on && c0 + jj <= lane && c0 + jj < N) v.dd(lane, c0 + jj) = row[jj];
            }
            double di = on ? v.d[lane] : 0.0;
            for (uint32_t c0 = 0; c0 < N; c0 += PB)
            {
#pragma unroll
                for (int jj = 0; jj < PB; jj++) row[jj] = (on && c0 + jj <= lane && c0 + jj < N) ? v.dd(lane, c0 + jj) : 0.0;
#pragma unroll
                for (int jj = 0; jj < PB; jj++)
                    if (c0 + jj < N)
                    {
                        const uint32_t j = c0 + jj;
                        const double yj  = reg_rdlane(di, (int)j) / reg_rdlane(row[jj], (int)j);
                        di               = (lane == j) ? yj : (lane > j ? rfma(-row[jj], yj, di) : di);
                    }
            }
            __syncthreads(); // the factor's columns are read across lanes from here on
            for (uint32_t c0 = ((N - 1) / PB) * PB; N > 0; c0 -= PB)
            {
#pragma unroll
                for (int jj = 0; jj < PB; jj++) row[jj] = (on && c0 + jj >= lane && c0 + jj < N) ? v.dd(c0 + jj, lane) : 0.0;
#pragma unroll
                for (int jj = PB - 1; jj >= 0; jj--)
                    if (c0 + jj < N)
                    {
                        const uint32_t j = c0 + jj;
                        const double zj  = reg_rdlane(di, (int)j) / reg_rdlane(row[jj], (int)j);
                        di               = (lane == j) ? zj : (lane < j ? rfma(-row[jj], zj, di) : di);
                    }
                if (c0 == 0) break;
            }
            if (on) v.d[lane] = di;
            __syncthreads();
        }

        /// LLT of the lower triangle of D (N x N) in place, then D z = d in place (oracle: cholesky_solve)
        template <int NT>
        __device__ void reg_cholesky_solve(const RegView &v, uint32_t N, uint32_t tid)
        {
            if constexpr (NT == 64)
            {
                if (N <= 64) return reg_cholesky_solve_wave(v, N, tid);
            }
            for (uint32_t j = 0; j < N; j++)
            {
                if (tid == 0)
                {
                    double sjj = v.dd(j, j);
                    for (uint32_t k = 0; k < j; k++) sjj = rfma(-v.dd(j, k), v.dd(j, k), sjj);
                    v.dd(j, j) = sqrt(sjj);
                }
                __syncthreads();
                const double ljj = v.dd(j, j);
                for (uint32_t i = j + 1 + tid; i < N; i += NT)
                {
                    double t = v.dd(i, j);
                    for (uint32_t k = 0; k < j; k++) t = rfma(-v.dd(i, k), v.dd(j, k), t);
                    v.dd(i, j) = t / ljj;
                }
                __syncthreads();
            }
            for (uint32_t j = 0; j < N; j++)
            {
                if (tid == 0) v.d[j] = v.d[j] / v.dd(j, j);
                __syncthreads();
                const double yj = v.d[j];
                for (uint32_t i = j + 1 + tid; i < N; i += NT) v.d[i] = rfma(-v.dd(i, j), yj, v.d[i]);
                __syncthreads();
            }
            for (uint32_t j = N; j--;)
            {
                if (tid == 0) v.d[j] = v.d[j] / v.dd(j, j);
                __syncthreads();
                const double zj = v.d[j];
                for (uint32_t i = tid; i < j; i += NT) v.d[i] = rfma(-v.dd(j, i), zj, v.d[i]);
                __syncthreads();
            }
        }

        /// d <- sym(D0) d (lower triangle stored)
        template <int NT>
        __device__ void reg_symv_lower_d0(const RegView &v, uint32_t N, uint32_t tid)
        {
            for (uint32_t i = tid; i < N; i += NT)
            {
                double acc = 0.0;
                for (uint32_t j = 0; j < N; j++) acc = rfma(i >= j ? v.d0(i, j) : v.d0(j, i), v.d[j], acc);
                v.out[i] = acc;
            }
            __syncthreads();
            for (uint32_t i = tid; i < N; i += NT) v.d[i] = v.out[i];
            __syncthreads();
        }

        /// entry e of the lower triangle (diagonal included) of an N x N matrix, column by column: e -> (i, j), i >= j.  The loops over a symmetric
        /// matrix's entries run over these N (N + 1) / 2 — not over N^2 with the upper half skipped, which idles half the lanes' trips
        __device__ __forceinline__ void reg_lower_entry(uint32_t e, uint32_t N, uint32_t &i, uint32_t &j)
        {
            const float twoNp1 = (float)(2u * N + 1u);
            uint32_t jj        = (uint32_t)((twoNp1 - sqrtf(twoNp1 * twoNp1 - 8.0f * (float)e)) * 0.5f);
            if (jj >= N) jj = N - 1u;
            while (jj > 0u && jj * N - (jj * (jj - 1u)) / 2u > e) jj--;                       // (the float estimate may be one off)
            while (jj + 1u < N && (jj + 1u) * N - ((jj + 1u) * jj) / 2u <= e) jj++;
            j = jj;
            i = jj + (e - (jj * N - (jj * (jj - 1u)) / 2u));
        }

        /// lower triangle of R^T R into D
        template <int NT>
        __device__ void reg_lower_RtR(const RegView &v, uint32_t F, uint32_t Fc, uint32_t rank, uint32_t tid)
        {
            for (uint32_t e = tid; e < rank * (rank + 1) / 2; e += NT)
            {
                uint32_t i, j;
                reg_lower_entry(e, rank, i, j);
                double acc = 0.0;
                for (uint32_t k = 0; k <= j; k++) acc = rfma(v.w(F + k, Fc + i), v.w(F + k, Fc + j), acc);
                v.dd(i, j) = acc;
            }
        }
        /// lower triangle of R R^T + T T^T into D
        template <int NT>
        __device__ void reg_lower_RRt_TTt(const RegView &v, uint32_t F, uint32_t Fc, uint32_t rank, uint32_t RC, uint32_t tid)
        {
            for (uint32_t e = tid; e < rank * (rank + 1) / 2; e += NT)
            {
                uint32_t i, j;
                reg_lower_entry(e, rank, i, j);
                double acc = 0.0;
                for (uint32_t k = i; k < rank; k++) acc = rfma(v.w(F + i, Fc + k), v.w(F + j, Fc + k), acc);
                double t = 0.0;
                for (uint32_t c = 0; c < RC; c++) t = rfma(v.w(F + i, Fc + rank + c), v.w(F + j, Fc + rank + c), t);
                v.dd(i, j) = acc + t;
            }
        }
        __device__ inline double reg_Rt_rhs(const RegView &v, uint32_t F, uint32_t Fc, uint32_t i)
        {
            double acc = 0.0;
            for (uint32_t k = 0; k <= i; k++) acc = rfma(v.w(F + k, Fc + i), v.w(F + k, v.n), acc);
            return acc;
        }
        __device__ inline double reg_R_times_d(const RegView &v, uint32_t F, uint32_t Fc, uint32_t rank, uint32_t i)
        {
            double acc = 0.0;
            for (uint32_t j = i; j < rank; j++) acc = rfma(v.w(F + i, Fc + j), v.d[j], acc);
            return acc;
        }
        template <int NT>
        __device__ void reg_copy_D_to_D0(const RegView &v, uint32_t N, uint32_t tid)
        {
            for (uint32_t e = tid; e < N * N; e += NT)
            {
                const uint32_t i = e % N, j = e / N;
                v.d0(i, j) = v.dd(i, j);
            }
            __syncthreads();
        }
        /// rhs of the level <- out[0..rank)
        template <int NT>
        __device__ void reg_store_rhs(const RegView &v, uint32_t F, uint32_t rank, const double *src, uint32_t tid)
        {
            __syncthreads();
            for (uint32_t i = tid; i < rank; i += NT) v.w(F + i, v.n) = src[i];
            __syncthreads();
        }

        template <int NT>
        __device__ void reg_tikhonov_1(const RegView &v0, uint32_t F, uint32_t Fc, uint32_t rank, uint32_t RC, double f, uint32_t tid)
        {
            const double mu   = f * f;
            const uint32_t m0 = Fc - v0.nf, N = RC + rank;
            const RegView v   = v0.with_order(N);
            reg_lower_RtR<NT>(v, F, Fc, rank, tid);
            for (uint32_t e = tid; e < RC * (RC + 1) / 2; e += NT) // Tk'*Tk (lower)
            {
                uint32_t a, b2;
                reg_lower_entry(e, RC, a, b2);
                double acc = 0.0;
                for (uint32_t k = 0; k < rank; k++) acc = rfma(v.w(F + k, Fc + rank + a), v.w(F + k, Fc + rank + b2), acc);
                v.dd(rank + a, rank + b2) = acc;
            }
            for (uint32_t e = tid; e < RC * rank; e += NT) // Tk'*triu(Rk)
            {
                const uint32_t a = e % RC, j = e / RC;
                double acc = 0.0;
                for (uint32_t k = 0; k <= j; k++) acc = rfma(v.w(F + k, Fc + rank + a), v.w(F + k, Fc + j), acc);
                v.dd(rank + a, j) = acc;
            }
            __syncthreads();
            for (uint32_t e = tid; e < N * (N + 1) / 2; e += NT) // += mu * up'*up ; + mu on the diagonal
            {
                uint32_t i, j;
                reg_lower_entry(e, N, i, j);
                double acc = 0.0;
                #pragma unroll 4
                for (uint32_t r = 0; r < m0; r++) acc = rfma(v.ns(r, Fc + i), v.ns(r, Fc + j), acc);
                double t = rfma(mu, acc, v.dd(i, j));
                if (i == j) t += mu;
                v.dd(i, j) = t;
            }
            for (uint32_t i = tid; i < N; i += NT)
            {
                double base;
                if (i < rank)
                    base = reg_Rt_rhs(v, F, Fc, i);
                else
                {
                    double acc = 0.0;
                    for (uint32_t k = 0; k < rank; k++) acc = rfma(v.w(F + k, Fc + i), v.w(F + k, v.n), acc);
                    base = acc;
                }
                double acc = 0.0;
                #pragma unroll 4
                for (uint32_t r = 0; r < m0; r++) acc = rfma(v.ns(r, Fc + i), v.ns(r, v.n), acc);
                v.d[i] = rfma(mu, acc, base);
            }
            __syncthreads();
            reg_cholesky_solve<NT>(v, N, tid);
            for (uint32_t i = tid; i < rank; i += NT)
            {
                double t = 0.0;
                for (uint32_t c = 0; c < RC; c++) t = rfma(v.w(F + i, Fc + rank + c), v.d[rank + c], t);
                v.out[i] = reg_R_times_d(v, F, Fc, rank, i) + t;
            }
            reg_store_rhs<NT>(v, F, rank, v.out, tid);
        }

        template <int NT>
        __device__ void reg_tikhonov_2(const RegView &v0, uint32_t F, uint32_t Fc, uint32_t rank, uint32_t RC, double f, uint32_t tid)
        {
            const double mu   = f * f;
            const uint32_t m0 = Fc - v0.nf, N = m0 + rank, Wd = RC + rank;
            const RegView v   = v0.with_order(N);
            reg_lower_RRt_TTt<NT>(v, F, Fc, rank, RC, tid);
            for (uint32_t e = tid; e < m0 * (m0 + 1) / 2; e += NT) // mu * up*up' (lower)
            {
                uint32_t s2, t;
                reg_lower_entry(e, m0, s2, t);
                double acc = 0.0;
                #pragma unroll 4
                for (uint32_t c = 0; c < Wd; c++) acc = rfma(v.ns(s2, Fc + c), v.ns(t, Fc + c), acc);
                v.dd(rank + s2, rank + t) = mu * acc;
            }
            for (uint32_t e = tid; e < m0 * rank; e += NT) // f * (up.leftCols(rank)*triu(Rk)' + up.rightCols(RC)*Tk')
            {
                const uint32_t s2 = e % m0, i = e / m0;
                double a1 = 0.0;
                #pragma unroll 4
                for (uint32_t c = i; c < rank; c++) a1 = rfma(v.ns(s2, Fc + c), v.w(F + i, Fc + c), a1);
                double a2 = 0.0;
                #pragma unroll 4
                for (uint32_t c = 0; c < RC; c++) a2 = rfma(v.ns(s2, Fc + rank + c), v.w(F + i, Fc + rank + c), a2);
                v.dd(rank + s2, i) = rfma(f, a2, f * a1);
            }
            __syncthreads();
            for (uint32_t i = tid; i < N; i += NT)
            {
                v.dd(i, i) += mu;
                v.d[i] = (i < rank) ? v.w(F + i, v.n) : f * v.ns(i - rank, v.n);
            }
            __syncthreads();
            reg_copy_D_to_D0<NT>(v, N, tid);
            reg_cholesky_solve<NT>(v, N, tid);
            for (uint32_t i = tid; i < N; i += NT) v.d0(i, i) -= mu;
            __syncthreads();
            reg_symv_lower_d0<NT>(v, N, tid);
            reg_store_rhs<NT>(v, F, rank, v.d, tid);
        }

        template <int NT>
        __device__ void reg_R(const RegView &v0, uint32_t F, uint32_t Fc, uint32_t rank, double f, bool with_z, uint32_t tid)
        {
            const double mu   = f * f;
            const uint32_t m0 = with_z ? Fc - v0.nf : 0;
            const RegView v   = v0.with_order(rank);
            REG_STAMP(v, 0, tid)
            reg_lower_RtR<NT>(v, F, Fc, rank, tid);
            __syncthreads();
            REG_STAMP(v, 1, tid)
            for (uint32_t e = tid; e < rank * (rank + 1) / 2; e += NT)
            {
                uint32_t i, j;
                reg_lower_entry(e, rank, i, j);
                double t = v.dd(i, j);
                if (with_z)
                {
                    double acc = 0.0;
                    #pragma unroll 4
                    for (uint32_t r = 0; r < m0; r++) acc = rfma(v.ns(r, Fc + i), v.ns(r, Fc + j), acc);
                    t = rfma(mu, acc, t);
                }
                if (i == j) t += mu;
                v.dd(i, j) = t;
            }
            for (uint32_t i = tid; i < rank; i += NT)
            {
                if (with_z)
                {
                    double acc = 0.0;
                    #pragma unroll 4
                    for (uint32_t r = 0; r < m0; r++) acc = rfma(v.ns(r, Fc + i), v.ns(r, v.n), acc);
                    v.d[i] = mu * acc + reg_Rt_rhs(v, F, Fc, i);
                }
                else
                    v.d[i] = reg_Rt_rhs(v, F, Fc, i);
            }
            __syncthreads();
            REG_STAMP(v, 2, tid)
            reg_cholesky_solve<NT>(v, rank, tid);
            REG_STAMP(v, 3, tid)
            for (uint32_t i = tid; i < rank; i += NT) v.out[i] = reg_R_times_d(v, F, Fc, rank, i);
            reg_store_rhs<NT>(v, F, rank, v.out, tid);
            REG_STAMP(v, 4, tid)
        }

        template <int NT>
        __device__ void reg_RT_NO_Z(const RegView &v0, uint32_t F, uint32_t Fc, uint32_t rank, uint32_t RC, double f, uint32_t tid)
        {
            const double mu = f * f;
            const RegView v = v0.with_order(rank);
            reg_lower_RRt_TTt<NT>(v, F, Fc, rank, RC, tid);
            __syncthreads();
            for (uint32_t i = tid; i < rank; i += NT)
            {
                v.dd(i, i) += mu;
                v.d[i] = v.w(F + i, v.n);
            }
            __syncthreads();
            reg_copy_D_to_D0<NT>(v, rank, tid);
            reg_cholesky_solve<NT>(v, rank, tid);
            for (uint32_t i = tid; i < rank; i += NT) v.d0(i, i) -= mu;
            __syncthreads();
            reg_symv_lower_d0<NT>(v, rank, tid);
            reg_store_rhs<NT>(v, F, rank, v.d, tid);
        }

        /// regularize_tikhonov_CG / regularize_RT_NO_Z_CG (lexlse.h:2256-2279, :2325-2347) with cg_tikhonov / cg_RT (:2370-2554): CGLS on
        /// [Rk Tk; f Sk; f I] x = [y; f s; 0] from x = 0; arithmetic order of the oracle's regularize_cg
        template <int NT>
        __device__ void reg_cg(const RegView &v, uint32_t F, uint32_t Fc, uint32_t rank, uint32_t RC, double f, bool with_z, uint32_t max_iter, uint32_t tid)
        {
            const uint32_t m0 = with_z ? Fc - v.nf : 0, N = rank + RC, n = v.n;
            double *x = v.cg, *r1 = x + n, *r2 = r1 + n, *r3 = r2 + n, *q1 = r3 + n, *q2 = q1 + n, *q3 = q2 + n, *sv = q3 + n, *pv = sv + n;
            auto T_times = [&](const double *vec, uint32_t i) {
                double t = 0.0;
                #pragma unroll 4
                for (uint32_t c = 0; c < RC; c++) t = rfma(v.w(F + i, Fc + rank + c), vec[rank + c], t);
                return t;
            };
            auto R_times = [&](const double *vec, uint32_t i) {
                double rr = 0.0;
                #pragma unroll 4
                for (uint32_t j = i; j < rank; j++) rr = rfma(v.w(F + i, Fc + j), vec[j], rr);
                return rr;
            };
            auto compute_s = [&]() {
                for (uint32_t i = tid; i < N; i += NT)
                {
                    double acc = 0.0;
                    #pragma unroll 4
                    for (uint32_t k = 0; k < m0; k++) acc = rfma(v.ns(k, Fc + i), r2[k], acc);
                    double sval = with_z ? (acc + r3[i]) * f : f * r3[i];
                    double add  = 0.0;
                    if (i < rank)
                        #pragma unroll 4
                        for (uint32_t k = 0; k <= i; k++) add = rfma(v.w(F + k, Fc + i), r1[k], add);
                    else
                        #pragma unroll 4
                        for (uint32_t k = 0; k < rank; k++) add = rfma(v.w(F + k, Fc + i), r1[k], add);
                    sv[i] = sval + add;
                }
                __syncthreads();
            };
            for (uint32_t i = tid; i < N; i += NT) x[i] = 0.0;
            __syncthreads();
            for (uint32_t i = tid; i < rank; i += NT)
            {
                double t = v.w(F + i, n) - T_times(x, i);
                t -= R_times(x, i);
                r1[i] = t;
            }
            for (uint32_t k = tid; k < m0; k += NT)
            {
                double acc = 0.0;
                #pragma unroll 4
                for (uint32_t i = 0; i < N; i++) acc = rfma(v.ns(k, Fc + i), x[i], acc);
                r2[k] = (v.ns(k, n) - acc) * f;
            }
            for (uint32_t i = tid; i < N; i += NT) r3[i] = -f * x[i];
            __syncthreads();
            compute_s();
            for (uint32_t i = tid; i < N; i += NT) pv[i] = sv[i];
            double gamma;
            if constexpr (NT == 64)
                gamma = reg_wave_sumsq(sv, N, tid, 0.0);
            else
            {
                if (tid == 0)
                {
                    double g = 0.0;
                    #pragma unroll 4
                    for (uint32_t i = 0; i < N; i++) g = rfma(sv[i], sv[i], g);
                    v.scal[1] = g;
                }
                __syncthreads();
                gamma = v.scal[1];
            }
            __syncthreads();
            uint32_t iter = 0;
            while (sqrt(gamma) > 1e-12 && iter < max_iter) // uniform: gamma is read from memory by every thread
            {
                for (uint32_t i = tid; i < rank; i += NT)
                {
                    double t = T_times(pv, i);
                    t += R_times(pv, i);
                    q1[i] = t;
                }
                for (uint32_t k = tid; k < m0; k += NT)
                {
                    double acc = 0.0;
                    #pragma unroll 4
                    for (uint32_t i = 0; i < N; i++) acc = rfma(v.ns(k, Fc + i), pv[i], acc);
                    q2[k] = acc * f;
                }
                for (uint32_t i = tid; i < N; i += NT) q3[i] = f * pv[i];
                __syncthreads();
                double alpha;
                if constexpr (NT == 64)
                    alpha = gamma / reg_wave_sumsq(q3, N, tid, reg_wave_sumsq(q2, m0, tid, reg_wave_sumsq(q1, rank, tid, 0.0)));
                else
                {
                    if (tid == 0)
                    {
                        double qq = 0.0;
                        #pragma unroll 4
                        for (uint32_t i = 0; i < rank; i++) qq = rfma(q1[i], q1[i], qq);
                        #pragma unroll 4
                        for (uint32_t k = 0; k < m0; k++) qq = rfma(q2[k], q2[k], qq);
                        #pragma unroll 4
                        for (uint32_t i = 0; i < N; i++) qq = rfma(q3[i], q3[i], qq);
                        v.scal[2] = gamma / qq;
                    }
                    __syncthreads();
                    alpha = v.scal[2];
                }
                for (uint32_t i = tid; i < N; i += NT)
                {
                    x[i]  = rfma(alpha, pv[i], x[i]);
                    r3[i] = rfma(-alpha, q3[i], r3[i]);
                }
                for (uint32_t i = tid; i < rank; i += NT) r1[i] = rfma(-alpha, q1[i], r1[i]);
                for (uint32_t k = tid; k < m0; k += NT) r2[k] = rfma(-alpha, q2[k], r2[k]);
                __syncthreads();
                compute_s();
                const double gamma_previous = gamma;
                if constexpr (NT == 64)
                    gamma = reg_wave_sumsq(sv, N, tid, 0.0);
                else
                {
                    if (tid == 0)
                    {
                        double g = 0.0;
                        #pragma unroll 4
                        for (uint32_t i = 0; i < N; i++) g = rfma(sv[i], sv[i], g);
                        v.scal[1] = g;
                    }
                    __syncthreads();
                    gamma = v.scal[1];
                }
                const double beta           = gamma / gamma_previous;
                for (uint32_t i = tid; i < N; i += NT) pv[i] = rfma(beta, pv[i], sv[i]);
                __syncthreads();
                iter++;
            }
            for (uint32_t i = tid; i < rank; i += NT) v.out[i] = R_times(x, i) + T_times(x, i);
            reg_store_rhs<NT>(v, F, rank, v.out, tid);
        }

        /// lexlse.h:2592-2625
        template <int NT>
        __device__ void reg_accumulate_nullspace(const RegView &v, uint32_t F, uint32_t Fc, uint32_t rank, uint32_t RC, uint32_t tid)
        {
            const uint32_t m0 = Fc - v.nf, rows = m0 + rank;
            for (uint32_t e = tid; e < rank * rank; e += NT)
            {
                const uint32_t i = e % rank, j = e / rank;
                v.ns(m0 + i, Fc + j) = (i == j) ? 1.0 : 0.0;
            }
            __syncthreads();
            if (rank == 0) return;
            for (uint32_t i = tid; i < rows; i += NT) // LeftBlock <- LeftBlock * R^-1 (row by row, reciprocal of the diagonal)
                for (uint32_t p = 0; p < rank; p++)
                {
                    double sv = v.ns(i, Fc + p);
                    #pragma unroll 4
                    for (uint32_t q = 0; q < p; q++) sv = rfma(-v.ns(i, Fc + q), v.w(F + q, Fc + p), sv);
                    v.ns(i, Fc + p) = sv * (1.0 / v.w(F + p, Fc + p));
                }
            __syncthreads();
            for (uint32_t e = tid; e < rows * (RC + 1); e += NT) // TrailingBlock -= LeftBlock * UpBlock (RHS column included)
            {
                const uint32_t i = e % rows, k = e / rows;
                double t = v.ns(i, Fc + rank + k);
                #pragma unroll 4
                for (uint32_t p = 0; p < rank; p++) t = rfma(-v.ns(i, Fc + p), v.w(F + p, Fc + rank + k), t);
                v.ns(i, Fc + rank + k) = t;
            }
            __syncthreads();
        }

        /// Eigen applyHouseholderOnTheLeft on one vector (oracle: apply_householder)
        __device__ inline void reg_apply_householder(const double *ess, double tau, double *x, uint32_t len)
        {
            if (len == 1)
                x[0] *= (1.0 - tau);
            else if (tau != 0.0)
            {
                double tmp = 0.0;
                for (uint32_t i = 1; i < len; i++) tmp = rfma(ess[i - 1], x[i], tmp);
                tmp += x[0];
                x[0] = rfma(-tau, tmp, x[0]);
                for (uint32_t i = 1; i < len; i++) x[i] = rfma(-(tau * ess[i - 1]), tmp, x[i]);
            }
        }

        /// lexlse.h:1774-1886 (oracle: regularize_tikhonov_1_test): regularize_tikhonov_1, then the residual of the regularized level
        /// and the regularized solution of the levels 0..ObjIndex (get_intermediate_x :2010-2071).  The by-products are a few hundred
        /// flops of strictly ordered chains: one thread.
        template <int NT>
        __device__ void reg_tikhonov_1_test(const RegView &v, const LseArgs &a, uint32_t b, const RegLevels &lv, uint32_t ObjIndex, uint32_t F, uint32_t Fc,
                                            uint32_t rank, uint32_t RC, double f, uint32_t tid)
        {
            reg_tikhonov_1<NT>(v, F, Fc, rank, RC, f, tid); // leaves the solution of the normal equations in v.d, ends with a barrier
            if (tid == 0)
            {
                const uint32_t n = v.n, N = RC + rank, dim = lv.dim;
                double *mu  = a.reg_mu + (size_t)b * reg_mu_doubles(n, a.nObj, a.cap);
                double *X   = mu + (size_t)ObjIndex * n;
                double *res = mu + 2 * (size_t)a.nObj * n;
                double *w   = res + a.cap;
                for (uint32_t i = 0; i < dim; i++) w[i] = i < rank ? v.w(F + i, n) : 0.0; // Q1 [R T] d - b (:1848-1854)
                for (uint32_t j = rank; j--;) reg_apply_householder(&v.w(F + j + 1, Fc + j), lv.hh[F + j], w + j, dim - j);
                for (uint32_t i = 0; i < dim; i++) res[F + i] = w[i] - res[F + i];

                for (uint32_t i = 0; i < N; i++) X[n - N + i] = v.d[i]; // :1857
                for (uint32_t i = 0; i < ObjIndex; i++)                 // :2026-2040
                {
                    const uint32_t Fi = lv.fr[i], Fci = lv.fc[i], ri = lv.rk[i];
                    for (uint32_t r = 0; r < ri; r++)
                    {
                        double acc = 0.0;
                        for (uint32_t c = 0; c < N; c++) acc = rfma(v.w(Fi + r, n - N + c), X[n - N + c], acc);
                        X[Fci + r] = v.w(Fi + r, n) - acc;
                    }
                }
                uint32_t acc_ranks = 0;
                for (uint32_t k = ObjIndex; k--;) // :2046-2070
                {
                    const uint32_t Fk = lv.fr[k], Fck = lv.fc[k], rk = lv.rk[k];
                    if (rk == 0) continue;
                    if (acc_ranks > 0)
                    {
                        const uint32_t c0 = lv.fc[k + 1];
                        for (uint32_t i = 0; i < rk; i++)
                        {
                            double s = X[Fck + i];
                            for (uint32_t j = 0; j < acc_ranks; j++) s = rfma(-v.w(Fk + i, c0 + j), X[c0 + j], s);
                            X[Fck + i] = s;
                        }
                    }
                    for (uint32_t j = rk; j--;)
                    {
                        X[Fck + j] = X[Fck + j] / v.w(Fk + j, Fck + j);
                        for (uint32_t i = 0; i < j; i++) X[Fck + i] = rfma(-v.w(Fk + i, Fck + j), X[Fck + j], X[Fck + i]);
                    }
                    acc_ranks += rk;
                }
                uint32_t total = rank; // :1863-1874 (ranks without the fixed variables, as the reference counts them)
                for (uint32_t k = 0; k < ObjIndex; k++) total += lv.rk[k];
                for (uint32_t k = total; k--;)
                {
                    const uint32_t j = lv.perm[k];
                    const double t   = X[k];
                    X[k]             = X[j];
                    X[j]             = t;
                }
            }
            __syncthreads();
        }

        /// dispatch of lexlse.h:277-395 for one level; called by every thread of the workgroup (uniform arguments)
        template <int NT>
        __device__ __noinline__ void regularize_level(const LseArgs &a, uint32_t b, const RegView &v, uint32_t ObjIndex, uint32_t F, uint32_t Fc, uint32_t rank, uint32_t RC,
                                         uint32_t tid, const RegLevels *lv = nullptr)
        {
            double f;
            if (a.reg_variable == 0.0) // lexlse.h:277-311: the constant factor — every thread reads it (no hand-off through memory)
                f = v.has_fk ? v.fk : a.reg_factor[(size_t)b * a.nObj + ObjIndex];
            else
            {
                if (tid == 0) // the conditioning-dependent factor
                {
                    const double fk = v.has_fk ? v.fk : a.reg_factor[(size_t)b * a.nObj + ObjIndex];
                    f               = 0.0;
                    if (rank > 0)
                    {
                        double ce = 0.0;
                        for (uint32_t i = 0; i < rank; i++)
                        {
                            v.out[i] = v.w(F + i, v.n);
                            ce       = rfma(v.out[i], v.out[i], ce);
                        }
                        for (uint32_t j = rank; j--;)
                        {
                            v.out[j] = v.out[j] / v.w(F + j, Fc + j);
                            for (uint32_t i = 0; i < j; i++) v.out[i] = rfma(-v.w(F + i, Fc + j), v.out[j], v.out[i]);
                        }
                        double q = 0.0;
                        for (uint32_t i = 0; i < rank; i++) q = rfma(v.out[i], v.out[i], q);
                        ce /= q;
                        const double eps = a.reg_variable;
                        if (ce < eps)
                        {
                            f = sqrt(1 - (ce * ce) / (eps * eps));
                            f *= fk;
                        }
                    }
                    v.scal[0] = f;
                }
                __syncthreads();
                f = v.scal[0];
            }
            const bool nonzero = !(fabs(f - 0.0) < 1e-15);
            REG_STAMP(v, 0, tid)
            switch (a.reg_type)
            {
            case 1: // REGULARIZATION_TIKHONOV
                if (nonzero)
                {
                    if (Fc + rank <= RC)
                        reg_tikhonov_2<NT>(v, F, Fc, rank, RC, f, tid);
                    else
                        reg_tikhonov_1<NT>(v, F, Fc, rank, RC, f, tid);
                }
                reg_accumulate_nullspace<NT>(v, F, Fc, rank, RC, tid);
                break;
            case 7: // REGULARIZATION_TIKHONOV_1 (experimental in the reference; the generic kernel passes the level lists)
                if (nonzero && lv) reg_tikhonov_1_test<NT>(v, a, b, *lv, ObjIndex, F, Fc, rank, RC, f, tid);
                reg_accumulate_nullspace<NT>(v, F, Fc, rank, RC, tid);
                break;
            case 8: // REGULARIZATION_TIKHONOV_2
                if (nonzero) reg_tikhonov_2<NT>(v, F, Fc, rank, RC, f, tid);
                reg_accumulate_nullspace<NT>(v, F, Fc, rank, RC, tid);
                break;
            case 2: // REGULARIZATION_TIKHONOV_CG
                if (nonzero) reg_cg<NT>(v, F, Fc, rank, RC, f, true, a.reg_cg_iters, tid);
                reg_accumulate_nullspace<NT>(v, F, Fc, rank, RC, tid);
                break;
            case 6: // REGULARIZATION_RT_NO_Z_CG
                if (nonzero) reg_cg<NT>(v, F, Fc, rank, RC, f, false, a.reg_cg_iters, tid);
                break;
            case 3: // REGULARIZATION_R
                if (nonzero) reg_R<NT>(v, F, Fc, rank, f, true, tid);
                reg_accumulate_nullspace<NT>(v, F, Fc, rank, RC, tid);
                break;
            case 4: // REGULARIZATION_R_NO_Z
                if (nonzero) reg_R<NT>(v, F, Fc, rank, f, false, tid);
                break;
            case 5: // REGULARIZATION_RT_NO_Z
                if (nonzero) reg_RT_NO_Z<NT>(v, F, Fc, rank, RC, f, tid);
                break;
            case 9: // REGULARIZATION_TEST (lexlse.h:2244)
                if (nonzero)
                    for (uint32_t i = tid; i < rank; i += NT) v.w(F + i, v.n) *= f;
                __syncthreads();
                break;
            default: break;
            }
            REG_STAMP(v, 5, tid)
        }
        template <int NT>
        __device__ void regularize_level(const LseArgs &a, uint32_t b, double *W, size_t ld, uint32_t nf, uint32_t ObjIndex, uint32_t F, uint32_t Fc,
                                         uint32_t rank, uint32_t RC, uint32_t tid, const RegLevels *lv = nullptr)
        {
            regularize_level<NT>(a, b, reg_view(a, b, W, ld, nf), ObjIndex, F, Fc, rank, RC, tid, lv);
        }
    } // namespace
} // namespace lexls
