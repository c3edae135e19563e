// Large-problem lexicographic-QR path (BASELINE configs[1]: n = 512, 4 levels x 256 rows): the matrix (4.2 MB) does not fit
// a CU's LDS, so the factorization is spread over the whole chip, ONE KERNEL LAUNCH PER STAGE instead of one persistent
// workgroup — no in-kernel grid barrier, no spinning: stages are ordered by the stream.
//
//   per level:   level_begin   (initial squared column norms, one lane per column)
//   per pivot:   pivot         (one workgroup: first-maximum search, fresh/tail norms, Householder scalars, column swap,
//                               essential part)                                                  [serial part of the pivot]
//                apply         (one workgroup per tile of TC trailing columns, tile staged in LDS with coalesced loads:
//                               ordered dot product + update per column, norm down-date)         [parallel part]
//   per level:   level_end, trsm (row-per-lane, multipliers in LDS), gemm (row-per-lane x TJ columns in registers)
//
// Same arithmetic contract as the other kernels (oracle/lexlse_oracle.h): every chain is evaluated in the same order, so the
// result is bit-identical to the oracle's.  Whether a stage has anything to do (rank break, columns exhausted) is decided on
// the device from a small per-problem state record; the host only reads `exhausted` back once per level to stop launching.
#include "lexls_kernels.h"
#include "lexls_launch.h"
#include "lqr_wave_common.h"

#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace lexls
{
    struct LargeState
    {
        uint32_t ColIndex, rank, exhausted, F, dim, Fc, last_id, cur, piv, row, R, degenerate, totalrank, stop_level;
        double tau, diag, den;
    };

    namespace
    {
#ifndef LEXLS_LARGE_TC
#define LEXLS_LARGE_TC 8
#endif
        constexpr int TC = LEXLS_LARGE_TC; // trailing columns per apply-workgroup
        constexpr int TJ = 8;  // trailing columns per lane in the Gauss update

        __device__ __forceinline__ bool skipped(const LseArgs &a, uint32_t b) { return a.skip && a.skip[b]; }

        __global__ __launch_bounds__(256) void large_init(LseArgs a, LargeState *st)
        {
            const uint32_t b = blockIdx.y;
            if (skipped(a, b)) return;
            const size_t ps  = (size_t)a.cap * (a.nVar + 1);
            const double *in = a.in + b * ps;
            double *W        = a.fac + b * ps;
            const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
            if (in != W)
                for (size_t e = tid; e < ps; e += nth) W[e] = in[e];
            for (size_t i = tid; i < a.cap; i += nth) a.hh[(size_t)b * a.cap + i] = 0.0;
            for (size_t i = tid; i < a.nVar; i += nth) a.perm[(size_t)b * a.nVar + i] = (uint32_t)i;
            if (tid == 0)
            {
                LargeState z = {};
                st[b]        = z;
            }
        }

        __global__ __launch_bounds__(64) void large_level_begin(LseArgs a, LargeState *st, double *norms, uint32_t level)
        {
            const uint32_t b = blockIdx.y;
            if (skipped(a, b)) return;
            const uint32_t n = a.nVar, cap = a.cap;
            const uint32_t *dims = a.dims + (size_t)b * a.nObj;
            uint32_t F = 0;
            for (uint32_t k = 0; k < level; k++) F += dims[k];
            const uint32_t dim = dims[level];
            LargeState *s      = st + b;
            const uint32_t c0  = s->ColIndex;
            const double *W    = a.fac + (size_t)b * cap * (n + 1);
            const uint32_t k   = blockIdx.x * 64 + threadIdx.x;
            if (k < n && k >= c0 && !s->exhausted)
            {
                double acc = 0.0;
                for (uint32_t i = 0; i < dim; i++)
                {
                    const double w = W[F + i + (size_t)k * cap];
                    acc            = dfma(w, w, acc);
                }
                norms[(size_t)b * n + k] = acc;
            }
            if (blockIdx.x == 0 && threadIdx.x == 0)
            {
                s->F    = F;
                s->dim  = dim;
                s->Fc   = c0;
                s->rank = 0;
            }
        }

        /// serial part of one pivot (lexlse.h:205-242): one workgroup per problem
#ifndef LEXLS_LARGE_NTP
#define LEXLS_LARGE_NTP 1024
#endif
        constexpr uint32_t NTP = LEXLS_LARGE_NTP; // threads of the one-workgroup pivot kernel
        __global__ __launch_bounds__(NTP) void large_pivot(LseArgs a, LargeState *st, double *norms_all, uint32_t level, uint32_t counter)
        {
            extern __shared__ double smem[];
            const uint32_t b = blockIdx.y, tid = threadIdx.x;
            if (skipped(a, b)) return;
            LargeState *s = st + b;
            if (s->exhausted || counter >= s->dim || s->stop_level == level + 1) return;
            const uint32_t n = a.nVar, cap = a.cap, nObj = a.nObj;
            const uint32_t *dims = a.dims + (size_t)b * nObj;
            uint32_t M           = 0;
            for (uint32_t k = 0; k < nObj; k++) M += dims[k];
            double *W      = a.fac + (size_t)b * cap * (n + 1);
            double *norms  = norms_all + (size_t)b * n;
            const uint32_t c = s->ColIndex, row = s->F + counter, R = s->dim - counter;

            double *colv    = smem;                 // R
            double *red_v   = colv + ((R + 1) & ~1u); // 1024
            uint32_t *red_i = reinterpret_cast<uint32_t *>(red_v + 1024);
            double *sc      = reinterpret_cast<double *>(red_i + 1024); // fresh, tail, tau, diag, den, flags

            // first maximum of the down-dated norms
            double bv   = -INFINITY;
            uint32_t bi = 0xffffffffu;
            for (uint32_t k = c + tid; k < n; k += NTP)
                if (norms[k] > bv)
                {
                    bv = norms[k];
                    bi = k;
                }
            // wave-level first maximum (DPP butterfly + ballot; ties go to the lowest column index), then 16 wave results through LDS
            {
                const double wm          = wave_max(bv);
                unsigned long long tied  = __ballot(bv == wm && bi != 0xffffffffu);
                uint32_t best            = 0xffffffffu;
                while (tied) // more than one lane only on exact ties
                {
                    const int l = (int)__builtin_ctzll(tied);
                    tied &= tied - 1;
                    const uint32_t idx = (uint32_t)__builtin_amdgcn_readlane((int)bi, l);
                    best               = idx < best ? idx : best;
                }
                if ((tid & 63u) == 0)
                {
                    red_v[tid >> 6] = wm;
                    red_i[tid >> 6] = best;
                }
            }
            __syncthreads();
            if (tid == 0)
            {
                double v0   = red_v[0];
                uint32_t i0 = red_i[0];
                for (int w = 1; w < (int)(NTP / 64); w++)
                {
                    const double v2   = red_v[w];
                    const uint32_t i2 = red_i[w];
                    if (i2 != 0xffffffffu && (i0 == 0xffffffffu || v2 > v0 || (v2 == v0 && i2 < i0)))
                    {
                        v0 = v2;
                        i0 = i2;
                    }
                }
                red_i[0] = i0;
            }
            __syncthreads();
            const uint32_t piv = red_i[0];

            // stage the pivot column (coalesced) and run the two ordered chains on two different waves
            for (uint32_t i = tid; i < R; i += NTP) colv[i] = W[row + i + (size_t)piv * cap];
            __syncthreads();
            if (tid == 0)
            {
                double f = 0.0;
#pragma unroll 16
                for (uint32_t i = 0; i < R; i++) f = dfma(colv[i], colv[i], f);
                sc[0] = f;
            }
            if (tid == 64)
            {
                double t = 0.0;
#pragma unroll 16
                for (uint32_t i = 1; i < R; i++) t = dfma(colv[i], colv[i], t);
                sc[1] = t;
            }
            __syncthreads();
            const double fresh = sc[0];
            if (fresh < a.tol) // rank test on the squared norm (lexlse.h:214)
            {
                if (tid == 0)
                {
                    norms[piv]    = fresh;
                    s->stop_level = level + 1;
                }
                return;
            }
            const double c0v = colv[0];
            double tau = 0.0, diag = c0v, den = 1.0;
            int degenerate = 0;
            if (R > 1)
            {
                const double tailSq = sc[1];
                if (tailSq <= DBL_MIN)
                    degenerate = 1;
                else
                {
                    double beta = sqrt(dfma(c0v, c0v, tailSq));
                    if (c0v >= 0.0) beta = -beta;
                    diag = beta;
                    den  = c0v - beta;
                    tau  = (beta - c0v) / beta;
                }
            }
            // column swap over ALL rows (lexlse.h:222-232) fused with writing beta / the essential part
            for (uint32_t i = tid; i < M; i += NTP)
            {
                const double a1 = W[i + (size_t)c * cap];
                const double a2 = (i >= row && i < row + R) ? colv[i - row] : W[i + (size_t)piv * cap];
                double newc     = a2;
                if (R > 1)
                {
                    if (i == row)
                        newc = diag;
                    else if (i > row && i < row + R)
                        newc = degenerate ? 0.0 : a2 / den;
                }
                W[i + (size_t)c * cap] = newc;
                if (piv != c) W[i + (size_t)piv * cap] = a1;
            }
            if (tid == 0)
            {
                norms[piv]                  = fresh;
                a.perm[(size_t)b * n + c]   = piv;
                const double t              = norms[c];
                norms[c]                    = norms[piv];
                norms[piv]                  = t;
                if (R > 1) a.hh[(size_t)b * cap + row] = tau;
                s->cur        = c;
                s->piv        = piv;
                s->row        = row;
                s->R          = R;
                s->tau        = tau;
                s->diag       = diag;
                s->den        = den;
                s->degenerate = degenerate;
                s->last_id    = (level << 16) | counter;
                s->ColIndex   = c + 1;
                s->rank       = s->rank + 1;
                if (c + 1 == n) s->exhausted = 1;
            }
        }

        /// parallel part of one pivot (lexlse.h:243-266): a tile of TC trailing columns per workgroup
        __global__ __launch_bounds__(256) void large_apply(LseArgs a, const LargeState *st, double *norms_all, uint32_t level, uint32_t counter)
        {
            extern __shared__ double smem[];
            const uint32_t b = blockIdx.y, tid = threadIdx.x;
            if (skipped(a, b)) return;
            const LargeState *s = st + b;
            if (s->last_id != ((level << 16) | counter) || s->stop_level == level + 1) return;
            const uint32_t n = a.nVar, cap = a.cap;
            const uint32_t c = s->cur, row = s->row, R = s->R;
            const uint32_t j0 = c + 1 + blockIdx.x * TC;
            if (j0 > n) return;
            const uint32_t nc = (n + 1 - j0 < (uint32_t)TC) ? n + 1 - j0 : TC;
            double *W         = a.fac + (size_t)b * cap * (n + 1);
            double *norms     = norms_all + (size_t)b * n;
            const double tau  = s->tau;
            const uint32_t ld = R | 1u; // odd: lane t walks column t of the tile without bank conflicts
            double *tile      = smem;   // TC * ld
            double *es        = tile + (size_t)TC * ld; // R - 1 (+1 pad)
            double *tmps      = es + R;                 // TC: the dot products of the tile's columns

#ifdef LEXLS_LARGE_STAMPS
            long long t0 = clock64(), t1 = t0, t2 = t0, t3 = t0;
#endif
            if (R > 1 && tau != 0.0)
            {
                for (uint32_t i = tid; i + 1 < R; i += 256) es[i] = W[row + 1 + i + (size_t)c * cap];
                for (uint32_t i = tid; i < R; i += 256) // all TC loads of a row in flight before the first LDS store
                {
                    double v[TC];
#pragma unroll
                    for (int t = 0; t < TC; t++) v[t] = ((uint32_t)t < nc) ? W[row + i + (size_t)(j0 + t) * cap] : 0.0;
#pragma unroll
                    for (int t = 0; t < TC; t++) tile[t * ld + i] = v[t];
                }
                __syncthreads();
#ifdef LEXLS_LARGE_STAMPS
                t1 = clock64();
#endif
                if (tid < nc)
                {
                    // the ordered dot product of lexlse.h:243-246 (applyHouseholderOnTheLeft): one chain per column, the only serial part
                    const double *__restrict__ col = tile + tid * ld;
                    const double *__restrict__ ev  = es;
                    double tmp                     = 0.0;
#pragma unroll 16
                    for (uint32_t i = 1; i < R; i++) tmp = dfma(ev[i - 1], col[i], tmp);
                    tmp += col[0];
                    tmps[tid]       = tmp;
                    const double c0 = dfma(-tau, tmp, col[0]);
                    if (j0 + tid < n && c + 1 < n) norms[j0 + tid] = dfma(-c0, c0, norms[j0 + tid]);
                }
                __syncthreads();
#ifdef LEXLS_LARGE_STAMPS
                t2 = clock64();
#endif
                // the update itself is independent per element: all threads, fused with the store of the tile
                for (uint32_t i = tid; i < R; i += 256)
                {
                    const double f = (i == 0) ? -tau : -(tau * es[i - 1]);
                    double v[TC];
#pragma unroll
                    for (int t = 0; t < TC; t++) v[t] = dfma(f, tmps[t], tile[t * ld + i]);
#pragma unroll
                    for (int t = 0; t < TC; t++)
                        if ((uint32_t)t < nc) W[row + i + (size_t)(j0 + t) * cap] = v[t];
                }
#ifdef LEXLS_LARGE_STAMPS
                __syncthreads();
                t3 = clock64();
                if (tid == 0 && blockIdx.x == 0)
                {
                    double *dbg = a.lambda + (size_t)b * (n + cap);
                    dbg[0] += (double)(t1 - t0);
                    dbg[1] += (double)(t2 - t1);
                    dbg[2] += (double)(t3 - t2);
                    dbg[3] += 1.0;
                }
#endif
            }
            else if (tid < nc && j0 + tid < n && c + 1 < n)
            {
                const double w  = W[row + (size_t)(j0 + tid) * cap];
                norms[j0 + tid] = dfma(-w, w, norms[j0 + tid]);
            }
        }

        __global__ void large_level_end(LseArgs a, LargeState *st, uint32_t level)
        {
            const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
            if (b >= a.batch || skipped(a, b)) return;
            LargeState *s                        = st + b;
            a.rank[(size_t)b * a.nObj + level]   = s->rank;
            a.fcol[(size_t)b * a.nObj + level]   = s->Fc;
            s->totalrank += s->rank;
            a.totalrank[b] = s->totalrank;
        }

        /// L <- A_left R^-1 for the rows below the level (lexlse.h:450-452): one row per lane, multipliers in LDS
        __global__ __launch_bounds__(64) void large_trsm(LseArgs a, const LargeState *st, uint32_t level)
        {
            extern __shared__ double smem[];
            const uint32_t b = blockIdx.y, lane = threadIdx.x;
            if (skipped(a, b)) return;
            const LargeState *s = st + b;
            const uint32_t rank = s->rank;
            if (rank == 0 || level + 1 >= a.nObj) return;
            const uint32_t n = a.nVar, cap = a.cap, nObj = a.nObj;
            const uint32_t *dims = a.dims + (size_t)b * nObj;
            uint32_t M           = 0;
            for (uint32_t k = 0; k < nObj; k++) M += dims[k];
            const uint32_t F = s->F, Fc = s->Fc, Fn = F + s->dim;
            const uint32_t gi = Fn + blockIdx.x * 64 + lane;
            const bool on     = gi < M;
            double *W         = a.fac + (size_t)b * cap * (n + 1);
            double *__restrict__ L    = smem;                     // rank x 64, L[q * 64 + lane]
            double *__restrict__ Rrow = smem + (size_t)rank * 64; // rank: row p of R, staged once per p
            for (uint32_t q = 0; q < rank; q++) L[q * 64 + lane] = on ? W[gi + (size_t)(Fc + q) * cap] : 0.0;
            for (uint32_t p = 0; p < rank; p++)
            {
                __syncthreads();
                for (uint32_t q = p + lane; q < rank; q += 64) Rrow[q] = W[F + p + (size_t)(Fc + q) * cap];
                __syncthreads();
                const double inv = 1.0 / Rrow[p];
                const double Lp  = L[p * 64 + lane] * inv;
                L[p * 64 + lane] = Lp;
                uint32_t q = p + 1;
                for (; q + 8 <= rank; q += 8) // chunks of 8: reads first, then the stores
                {
                    double r8[8], l8[8];
#pragma unroll
                    for (int k = 0; k < 8; k++)
                    {
                        r8[k] = Rrow[q + k];
                        l8[k] = L[(q + k) * 64 + lane];
                    }
#pragma unroll
                    for (int k = 0; k < 8; k++) L[(q + k) * 64 + lane] = dfma(-Lp, r8[k], l8[k]);
                }
                for (; q < rank; q++) L[q * 64 + lane] = dfma(-Lp, Rrow[q], L[q * 64 + lane]);
            }
            if (on)
                for (uint32_t q = 0; q < rank; q++) W[gi + (size_t)(Fc + q) * cap] = L[q * 64 + lane];
        }

        /// L <- A_left R^-1, column per thread (level dims <= 1024): thread q owns column q of the block for TRB rows held in registers.
        /// Step p: thread p finalises L_.p = acc * (1/R_pp) and posts it in LDS; every thread q > p absorbs -L_.p * R[p][q].  Each element
        /// sees the same ordered chain as in the row-per-lane form above (p ascending, then the product with the reciprocal), so the result
        /// is bit-identical — but the chain of one row is spread over `rank` threads instead of being walked by one, and a thread streams
        /// its own (contiguous) column of R in chunks of TCH.
        constexpr int TRB = 8, TCH = 16;
        __global__ __launch_bounds__(1024) void large_trsm_cols(LseArgs a, const LargeState *st, uint32_t level)
        {
            __shared__ double Lb[2][TRB];
            const uint32_t b = blockIdx.y, tid = threadIdx.x;
            if (skipped(a, b)) return;
            const LargeState *s = st + b;
            const uint32_t rank = s->rank;
            if (rank == 0 || level + 1 >= a.nObj) return;
            const uint32_t n = a.nVar, cap = a.cap, nObj = a.nObj;
            const uint32_t *dims = a.dims + (size_t)b * nObj;
            uint32_t M           = 0;
            for (uint32_t k = 0; k < nObj; k++) M += dims[k];
            const uint32_t F = s->F, Fc = s->Fc, Fn = F + s->dim;
            const uint32_t r0 = Fn + blockIdx.x * TRB;
            if (r0 >= M) return; // uniform per workgroup
            double *W         = a.fac + (size_t)b * cap * (n + 1);
            const bool mine   = tid < rank;
            double *col       = W + (size_t)(Fc + (mine ? tid : 0)) * cap;
            double acc[TRB];
#pragma unroll
            for (int i = 0; i < TRB; i++) acc[i] = (mine && r0 + i < M) ? col[r0 + i] : 0.0;
            for (uint32_t p0 = 0; p0 < rank; p0 += TCH)
            {
                double rbuf[TCH]; // R[p0 .. p0+TCH)[tid]: this thread's column of R, contiguous in memory
#pragma unroll
                for (int j = 0; j < TCH; j++) rbuf[j] = (mine && p0 + j < rank && p0 + j <= tid) ? col[F + p0 + j] : 0.0;
#pragma unroll
                for (int j = 0; j < TCH; j++)
                {
                    const uint32_t p = p0 + j;
                    if (p < rank) // uniform
                    {
                        const int buf = (int)(p & 1u);
                        if (tid == p)
                        {
                            const double inv = 1.0 / rbuf[j];
#pragma unroll
                            for (int i = 0; i < TRB; i++)
                            {
                                acc[i]     = acc[i] * inv;
                                Lb[buf][i] = acc[i];
                            }
                        }
                        __syncthreads();
                        if (mine && tid > p)
                        {
#pragma unroll
                            for (int i = 0; i < TRB; i++) acc[i] = dfma(-Lb[buf][i], rbuf[j], acc[i]);
                        }
                    }
                }
            }
            if (mine)
            {
#pragma unroll
                for (int i = 0; i < TRB; i++)
                    if (r0 + i < M) col[r0 + i] = acc[i];
            }
        }

        /// Trailing -= L * Up (lexlse.h:454-469): one row per lane, TJ trailing columns per lane in registers
        __global__ __launch_bounds__(64) void large_gemm(LseArgs a, const LargeState *st, uint32_t level)
        {
            const uint32_t b = blockIdx.z, lane = threadIdx.x;
            if (skipped(a, b)) return;
            const LargeState *s = st + b;
            const uint32_t rank = s->rank;
            if (rank == 0 || level + 1 >= a.nObj) return;
            const uint32_t n = a.nVar, cap = a.cap, nObj = a.nObj;
            const uint32_t *dims = a.dims + (size_t)b * nObj;
            uint32_t M           = 0;
            for (uint32_t k = 0; k < nObj; k++) M += dims[k];
            const uint32_t F = s->F, Fc = s->Fc, Fn = F + s->dim, c = s->ColIndex;
            const uint32_t gi = Fn + blockIdx.x * 64 + lane;
            const uint32_t j0 = c + blockIdx.y * TJ;
            if (j0 > n) return;
            const bool on = gi < M;
            double *W     = a.fac + (size_t)b * cap * (n + 1);
            double acc[TJ];
#pragma unroll
            for (int t = 0; t < TJ; t++) acc[t] = (on && j0 + t <= n) ? W[gi + (size_t)(j0 + t) * cap] : 0.0;
#pragma unroll 4
            for (uint32_t p = 0; p < rank; p++)
            {
                const double l = on ? W[gi + (size_t)(Fc + p) * cap] : 0.0;
#pragma unroll
                for (int t = 0; t < TJ; t++)
                {
                    const double u = (j0 + t <= n) ? W[F + p + (size_t)(j0 + t) * cap] : 0.0; // wave-uniform
                    acc[t]         = dfma(-l, u, acc[t]);
                }
            }
#pragma unroll
            for (int t = 0; t < TJ; t++)
                if (on && j0 + t <= n) W[gi + (size_t)(j0 + t) * cap] = acc[t];
        }

        __global__ void large_finish(LseArgs a, const LargeState *st)
        {
            const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
            if (b >= a.batch || skipped(a, b)) return;
            // columns beyond TotalRank keep the identity permutation; levels after exhaustion already carry first_col = n
            a.totalrank[b] = st[b].totalrank;
        }
    } // namespace

    // =================================================================================================================
    // FAST large path (default for problems beyond one CU's LDS; the multi-launch path above stays as the bit-exact reference
    // behind lexls_lse_set_kernel_policy(h, 5)).
    //
    // north_star's contract for this path: pivots / ranks exact, x and residuals within 1e-10 — the 256-deep ORDERED chains of
    // the bit-exact path (one per pivot for the fresh norm, one per trailing column for the reflector's dot product: ~1 us each,
    // two dependent launches per pivot) are what costs 6.6 ms on configs[1].  Here
    //   * ONE launch per pivot: every workgroup finds the pivot itself (first maximum of the down-dated norms by position — 4 KB of
    //     L2 reads and a wave-level reduction, the same in every workgroup), forms the reflector from the pivot column with tree
    //     sums, applies it to its own tile of trailing columns (two columns per wavefront, rows across lanes) and down-dates
    //     their norms.  Nothing a launch reads is written by the same launch: norms, the position map and the state record are
    //     double-buffered by step parity; the pivot column (beta, essential part) goes to a side buffer.
    //   * columns are not moved during a level (position map, like the wave kernels); when the level ends ONE pass writes the
    //     whole matrix in the level's final column order into the second factor buffer (the reference swaps whole columns,
    //     lexlse.h:225) together with beta / the essential parts, so the Gauss step sees the layout of lexlse.h:431-471.
    //   * Gauss step: the column-per-thread TRSM above, then the trailing update  T -= L U  on the matrix cores
    //     (v_mfma_f64_16x16x4_f64, LDS-staged 64 x 64 x 16 tiles).  The instruction accumulates its four products in ascending
    //     k as fused multiply-adds starting from C (scripts/ubench/mfma_order.hip), i.e. exactly the contract's chain: this
    //     kernel is bit-identical to large_gemm and serves the bit-exact path as well.
    // =================================================================================================================
    namespace
    {
        typedef double v4f64 __attribute__((ext_vector_type(4)));

        template <int CTRL>
        __device__ __forceinline__ double dpp_add(double v)
        {
            const int lo2 = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
            const int hi2 = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
            return v + __hiloint2double(hi2, lo2);
        }
        __device__ __forceinline__ double wave_vmax(double x, double y)
        {
            double r;
            asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
            return r;
        }
        /// sum over the 64 lanes in a FIXED tree order (the same in every workgroup), wave-uniform result
        __device__ __forceinline__ double wave_sum(double v)
        {
            v = dpp_add<0xB1>(v);
            v = dpp_add<0x4E>(v);
            v = dpp_add<0x141>(v);
            v = dpp_add<0x140>(v);
            return (rdlane(v, 0) + rdlane(v, 16)) + (rdlane(v, 32) + rdlane(v, 48));
        }

#ifndef LEXLS_FAST_NW
#define LEXLS_FAST_NW 4
#endif
#ifndef LEXLS_FAST_CPW
#define LEXLS_FAST_CPW 1
#endif
        constexpr int FNW = LEXLS_FAST_NW, FNT = 64 * FNW; // wavefronts / threads per workgroup of the step kernel
        constexpr int FCPW = LEXLS_FAST_CPW;                // columns per wavefront
        constexpr int FTC  = FNW * FCPW;                    // columns per workgroup
        constexpr int FRC = 4;         // rows a lane keeps in registers between the dot product and the update (R <= 64 * FRC)

        struct FastBuffers
        {
            double *W[2];          // factor-sized work buffers, W[0] == a.fac
            double *norms[2];      // batch x n
            uint32_t *pos[2];      // batch x (n + 1): position of each physical column (entry n = RHS)
            LargeState *st[2];     // batch
            double *E;             // batch x eld x eld: essential parts, row q = pivot q of the level
            uint32_t eld;          // largest level dimension of the batch
            double *D;             // batch x n: beta by pivot position
        };

        __global__ __launch_bounds__(256) void fast_level_begin(LseArgs a, FastBuffers fb, uint32_t cur, uint32_t pp, uint32_t level)
        {
            const uint32_t b = blockIdx.y, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
            if (skipped(a, b)) return;
            const uint32_t n = a.nVar, cap = a.cap;
            const uint32_t *dims = a.dims + (size_t)b * a.nObj;
            uint32_t F = 0;
            for (uint32_t k = 0; k < level; k++) F += dims[k];
            const uint32_t dim = dims[level];
            LargeState *s      = fb.st[pp] + b;
            const uint32_t c0  = s->ColIndex;
            const double *W    = fb.W[cur] + (size_t)b * cap * (n + 1);
            const uint32_t k   = blockIdx.x * 4 + wave; // one column per wavefront
            if (k <= n)
            {
                if (lane == 0) fb.pos[pp][(size_t)b * (n + 1) + k] = k; // the buffer is in position order when a level starts
                if (k < n && k >= c0 && !s->exhausted)
                {
                    double acc = 0.0;
                    for (uint32_t i = lane; i < dim; i += 64)
                    {
                        const double w = W[F + i + (size_t)k * cap];
                        acc            = dfma(w, w, acc);
                    }
                    acc = wave_sum(acc);
                    if (lane == 0) fb.norms[pp][(size_t)b * n + k] = acc;
                }
            }
            __syncthreads(); // the state record is read above by every thread of this workgroup before thread 0 of block 0 rewrites it
            if (blockIdx.x == 0 && tid == 0)
            {
                s->F          = F;
                s->dim        = dim;
                s->Fc         = c0;
                s->rank       = 0;
                s->stop_level = 0;
                s->last_id    = pp; // which of the two position maps is current (steps that find nothing to do do not copy the maps forward)
            }
        }

        /// one pivot of the level: search, reflector, application to this workgroup's tile, norm down-date
        __global__ __launch_bounds__(FNT) void fast_step(LseArgs a, FastBuffers fb, uint32_t cur, uint32_t pin, uint32_t counter)
        {
            extern __shared__ double smem[];
            __shared__ double red_v[FNW];
            __shared__ uint32_t red_p[FNW], red_i[FNW];
            __shared__ double sums[2 * FNW];
            const uint32_t b = blockIdx.y, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
            if (skipped(a, b)) return;
            const uint32_t pout = pin ^ 1u;
            const uint32_t n = a.nVar, cap = a.cap;
            const double *norms_in = fb.norms[pin] + (size_t)b * n;
            double *norms_out      = fb.norms[pout] + (size_t)b * n;
            const uint32_t *pos_in = fb.pos[pin] + (size_t)b * (n + 1);
            uint32_t *pos_out      = fb.pos[pout] + (size_t)b * (n + 1);
            // Everything whose ADDRESS does not depend on the pivot is requested up front, so that a step is two dependent round trips to
            // L2 / memory (state + norms + positions + own tile, then the pivot column) instead of four: the search candidates of this thread
            constexpr int NCAND = 1024 / FNT; // candidates per thread in registers (n <= 1024); beyond that the loop below reads again
            double cv[NCAND];
            uint32_t cp[NCAND];
#pragma unroll
            for (int u = 0; u < NCAND; u++)
            {
                const uint32_t k = tid + (uint32_t)FNT * u;
                cv[u]            = (k < n) ? norms_in[k] : -1.0;
                cp[u]            = (k < n) ? pos_in[k] : 0u;
            }
            const LargeState s = fb.st[pin][b];
            LargeState *so     = fb.st[pout] + b;
            const bool owner   = blockIdx.x == 0;
            if (s.exhausted || s.stop_level || counter >= s.dim)
            {
                if (owner && tid == 0) *so = s;
                return;
            }
            double *W        = fb.W[cur] + (size_t)b * cap * (n + 1);
            const uint32_t c = s.ColIndex, row = s.F + counter, R = s.dim - counter;
            double *colv = smem;     // R: the pivot column
            double *es   = smem + R; // R: essential part (es[i], i >= 1)

            // ... and this wavefront's two columns of the tile (rows across the lanes)
            double keep[FCPW][FRC];
#pragma unroll
            for (int h = 0; h < FCPW; h++)
            {
                const uint32_t j = blockIdx.x * FTC + wave * FCPW + h;
#pragma unroll
                for (int u = 0; u < FRC; u++)
                {
                    const uint32_t i = lane + 64u * u;
                    keep[h][u]       = (j <= n && i < R) ? W[row + i + (size_t)j * cap] : 0.0;
                }
            }

            // ---- first maximum, by position, of the down-dated norms (lexlse.h:205-206): every workgroup for itself ----
            double bv   = -1.0;
            uint32_t bp = 0xffffffffu, bi = 0;
#pragma unroll
            for (int u = 0; u < NCAND; u++)
            {
                const uint32_t k = tid + (uint32_t)FNT * u;
                if (k < n && cp[u] >= c && (cv[u] > bv || (cv[u] == bv && cp[u] < bp)))
                {
                    bv = cv[u];
                    bp = cp[u];
                    bi = k;
                }
            }
            for (uint32_t k = tid + (uint32_t)FNT * NCAND; k < n; k += FNT)
            {
                const uint32_t p = pos_in[k];
                if (p >= c)
                {
                    const double v = norms_in[k];
                    if (v > bv || (v == bv && p < bp))
                    {
                        bv = v;
                        bp = p;
                        bi = k;
                    }
                }
            }
            {
                const double wm         = wave_max(bv);
                unsigned long long tied = __ballot(bv == wm && bp != 0xffffffffu);
                uint32_t bestp = 0xffffffffu, besti = 0;
                while (tied) // more than one lane only on exact ties
                {
                    const int l = (int)__builtin_ctzll(tied);
                    tied &= tied - 1;
                    const uint32_t p2 = (uint32_t)__builtin_amdgcn_readlane((int)bp, l);
                    const uint32_t i2 = (uint32_t)__builtin_amdgcn_readlane((int)bi, l);
                    if (p2 < bestp)
                    {
                        bestp = p2;
                        besti = i2;
                    }
                }
                if (lane == 0)
                {
                    red_v[wave] = wm;
                    red_p[wave] = bestp;
                    red_i[wave] = besti;
                }
            }
            __syncthreads();
            double v0   = red_v[0];
            uint32_t p0 = red_p[0], piv = red_i[0];
#pragma unroll
            for (int w = 1; w < FNW; w++)
            {
                const double v2   = red_v[w];
                const uint32_t p2 = red_p[w];
                if (p2 != 0xffffffffu && (p0 == 0xffffffffu || v2 > v0 || (v2 == v0 && p2 < p0)))
                {
                    v0  = v2;
                    p0  = p2;
                    piv = red_i[w];
                }
            }
            const uint32_t ppos = p0; // position of the pivot column before the swap = column_permutations entry

            // ---- the pivot column, its fresh norm and tail norm (lexlse.h:210-211, :241) ----
            double fr = 0.0, tl = 0.0;
            for (uint32_t i = tid; i < R; i += FNT)
            {
                const double w = W[row + i + (size_t)piv * cap];
                colv[i]        = w;
                fr             = dfma(w, w, fr);
                if (i > 0) tl = dfma(w, w, tl);
            }
            fr = wave_sum(fr);
            tl = wave_sum(tl);
            if (lane == 0)
            {
                sums[wave]     = fr;
                sums[FNW + wave] = tl;
            }
            __syncthreads();
            double fresh = sums[0], tailSq = sums[FNW]; // fixed order: the same in every workgroup
#pragma unroll
            for (int w = 1; w < FNW; w++)
            {
                fresh += sums[w];
                tailSq += sums[FNW + w];
            }
            if (fresh < a.tol) // rank test on the squared norm (lexlse.h:214): the level ends here
            {
                if (owner && tid == 0)
                {
                    *so            = s;
                    so->stop_level = 1;
                }
                return;
            }
            const double c0v = colv[0];
            double tau = 0.0, diag = c0v, den = 1.0;
            bool degenerate = true;
            if (R > 1 && !(tailSq <= DBL_MIN))
            {
                degenerate  = false;
                double beta = sqrt(dfma(c0v, c0v, tailSq));
                if (c0v >= 0.0) beta = -beta;
                diag = beta;
                den  = c0v - beta;
                tau  = (beta - c0v) / beta;
            }
            for (uint32_t i = 1 + tid; i < R; i += FNT) es[i] = degenerate ? 0.0 : colv[i] / den;
            __syncthreads();

            // ---- this workgroup's tile: two columns per wavefront, rows across the lanes ----
            const uint32_t posf = ppos;
#pragma unroll
            for (int h = 0; h < FCPW; h++)
            {
                const uint32_t j = blockIdx.x * FTC + wave * FCPW + h;
                if (j > n) continue; // wave-uniform
                uint32_t pj = (j == n) ? n : pos_in[j];
                // the swap of lexlse.h:222-232 on the position map
                uint32_t pnew = pj;
                if (j < n)
                {
                    if (j == piv)
                        pnew = c;
                    else if (pj == c)
                        pnew = posf;
                }
                if (lane == 0) pos_out[j] = pnew;
                const bool trailing = (j == n) || pnew > c;
                if (!trailing)
                {
                    if (lane == 0 && j < n) norms_out[j] = norms_in[j];
                    continue;
                }
                double *col = W + row + (size_t)j * cap;
                double a0n;
                const double a0 = rdlane(keep[h][0], 0);
                if (tau != 0.0)
                {
                    double part = 0.0;
#pragma unroll
                    for (int u = 0; u < FRC; u++)
                    {
                        const uint32_t i = lane + 64u * u;
                        if (i >= 1 && i < R) part = dfma(es[i], keep[h][u], part);
                    }
                    for (uint32_t i = lane + 64u * FRC; i < R; i += 64) part = dfma(es[i], col[i], part);
                    const double tmp = wave_sum(part) + a0; // applyHouseholderOnTheLeft (lexlse.h:243-246)
                    const double nt  = -tau;
                    a0n              = dfma(nt, tmp, a0);
#pragma unroll
                    for (int u = 0; u < FRC; u++)
                    {
                        const uint32_t i = lane + 64u * u;
                        if (i < R) col[i] = (i == 0) ? a0n : dfma(es[i] * nt, tmp, keep[h][u]);
                    }
                    for (uint32_t i = lane + 64u * FRC; i < R; i += 64) col[i] = dfma(es[i] * nt, tmp, col[i]);
                }
                else
                    a0n = a0;
                if (lane == 0 && j < n) norms_out[j] = dfma(-a0n, a0n, norms_in[j]); // lexlse.h:262-266
            }

            // ---- bookkeeping of the pivot: one workgroup ----
            if (owner)
            {
                double *E = fb.E + ((size_t)b * fb.eld + counter) * fb.eld; // row `counter` of this level's essential parts
                for (uint32_t i = 1 + tid; i < R; i += FNT) E[i] = es[i];
                if (tid == 0)
                {
                    fb.D[(size_t)b * n + c]     = diag;
                    a.perm[(size_t)b * n + c]   = ppos;
                    if (R > 1) a.hh[(size_t)b * cap + row] = tau;
                    *so            = s;
                    so->last_id    = pout;
                    so->ColIndex   = c + 1;
                    so->rank       = s.rank + 1;
                    so->exhausted  = (c + 1 == n) ? 1u : 0u;
                }
            }
        }

        // -----------------------------------------------------------------------------------------------------------------
        // The pivots of ONE level inside ONE launch (single problems: BASELINE configs[1]).  fast_step pays ~7 us per pivot: the floor of
        // a launch that depends on its predecessor (~3 us) plus two dependent round trips to data other XCDs wrote.  Here the level's
        // trailing matrix stays in the LDS of G workgroups (4 columns each: one per wavefront) for the whole level and a pivot costs ONE hand-off:
        //   every workgroup publishes a 16-byte record {largest down-dated norm of its columns, that column's position, tag | index} AND that
        //   column's remaining rows (its candidate for the pivot column), polls the G records of the pivot, picks the winner (first maximum
        //   by position — the same in every workgroup), reads the winner's column, forms the reflector itself and updates its own tile.
        //   Buffers alternate by pivot parity: a workgroup can only be one step ahead.
        // Hand-off protocol (MI355X_MICROARCH.md: data-tagged granules, the cheapest hand-off of its price list): every published 16 bytes —
        // a record, or one column entry {value, tag} — are ONE sc0 sc1 store carrying the pivot's tag and are read by ONE sc0 sc1 load that is
        // taken when the tag matches; nothing is drained, ordered or counted (the first version of this kernel drained the column stores,
        // added to an agent-scope counter and polled it: 6.4 us per pivot, no faster than a launch per pivot).  What made the difference,
        // measured with the stamps below: (1) no counter — 129 atomic adds to one address serialise; (2) records 256 bytes apart — 129
        // workgroups polling 129 records that share sixteen lines made one memory channel the bottleneck (2.80 -> 2.49 ms); (3) four polls per
        // lane in flight; (4) 4 columns per workgroup instead of 8 (the tile update is on the critical path).  Round 3 (3.5 -> 3.2 us per
        // pivot, rocprofv3: 816 us per 256-pivot level): what the next record depends on no longer goes through LDS — every wavefront keeps the
        // positions and down-dated norms of its own columns in registers, the wavefronts exchange their best through ONE 16-byte slot each and
        // one barrier, and pick by a maximum tree + a minimum tree over packed {position, column} keys (the scan of the workgroup's columns by
        // every thread cost 370 cycles per column, a compare-and-select chain 150 per wavefront); the owner of a candidate column ships its
        // fresh / tail squared norms behind the column (two more granules), so the readers need no reduction of their own; a thread's
        // column granule and the two sums are requested together (one wait); the reflector's scalars come from v_rsq_f64 / v_rcp_f64 +
        // Newton steps instead of three division / square-root sequences.  Per pivot now (scripts/persist_stamps.py, shader cycles at
        // ~2.3 GHz): records 2500 (1.2 polls of 129 records each), winner's column 1300, dot + scalars + pivot row 1400, exchange between the
        // wavefronts + record 800; column update, publication and bookkeeping (1100) overlap the others' waiting.
        // Tried and not kept as defaults (LEXLS_PERSIST_FORM="nw,cpw", LEXLS_LARGE_ONE_XCD=1): all workgroups on ONE XCD so that hand-offs stay in
        // its L2 (a plain store + sc1 load round trip is 1040 cycles there against 1800 across XCDs, scripts/ubench/xcd_pingpong.hip; an
        // all-to-all exchange of 32 workgroups 0.49 us against 1.0 us, scripts/ubench/xcd_exchange.hip) — but 129 workgroups then share 32 CUs,
        // the slowest one is 1-2 polls late every pivot, and the level is slower (2.26-2.6 ms against 2.16); 8 or 16 wavefronts per workgroup
        // (fewer records to poll, longer barriers: equal or slower); 2 or 4 columns per wavefront (+350 cycles per column on the chain).
        // The tags restart with every level, so the records and the column granules are cleared in front of every launch.  Every spin is bounded: a workgroup that gives up raises `abort`, which
        // every spin watches — the launch then ends WITHOUT committing anything and the host redoes the level with a launch per pivot.
        // -----------------------------------------------------------------------------------------------------------------
        struct PersistCtl
        {
            uint32_t arrive; // monotonic over the pivots of the level
            uint32_t abort;
            uint32_t done;   // workgroups that have finished every pivot of the level without giving up: the commit waits for all G
            uint32_t pad[13];
        };
        struct PersistCand
        {
            double norm;
            uint32_t pos, idx;
        };
        /// records per mailbox row (a reader's row starts on a 256-byte boundary)
        __host__ __device__ inline size_t persist_mailbox_stride(uint32_t G) { return ((size_t)G + 15u) & ~(size_t)15u; }
        /// bytes of the two (pivot parity) sets of G mailbox rows
        inline size_t persist_mailbox_bytes(uint32_t G) { return 2 * (size_t)G * persist_mailbox_stride(G) * sizeof(PersistCand); }
        __device__ __forceinline__ void st_sc1(double *p, double v)
        {
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __device__ __forceinline__ double ld_sc1(const double *p)
        {
            return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        constexpr int PTC_MIN = 4; // fewest columns per workgroup (= wavefronts per workgroup) of the forms below: sizes the workspace
        /// 16-byte hand-off accesses: a record {norm, pos, tag | column} travels as ONE store and is read as ONE load (observed untorn on gfx950,
        /// MI355X_MICROARCH.md "R2's granule"); payload columns go one {value, tag} per lane.
        /// L2LOCAL = false: writers and readers on any XCD — system-scope store and load (sc0 sc1), served by memory.
        /// L2LOCAL = true: ALL workgroups of the launch sit on ONE XCD and share its L2 — a plain store leaves the line in that L2 and an
        /// agent-scope (sc1) load, which misses the CU's vector cache, is served by it: 0.49 us per all-to-all exchange of 32 workgroups
        /// against 1.0 us across XCDs (scripts/ubench/xcd_exchange.hip; sc0 / unscoped loads never see the store: they hit the vector cache)
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        template <bool L2LOCAL> __device__ __forceinline__ void st16_x(void *p, u32x4 v)
        {
#ifndef LEXLS_ONEXCD_STORE
#define LEXLS_ONEXCD_STORE 0
#endif
            if (L2LOCAL)
            {
                if (LEXLS_ONEXCD_STORE == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
                if (LEXLS_ONEXCD_STORE == 1) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
                if (LEXLS_ONEXCD_STORE == 2) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
                if (LEXLS_ONEXCD_STORE == 3) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory");
                if (LEXLS_ONEXCD_STORE == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
                if (LEXLS_ONEXCD_STORE == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
            }
            else
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
        }
        /// N loads in flight together, ONE wait — inside ONE asm statement: an output register of a load that is still in flight must not be
        /// visible to the compiler (it may copy it — a phi of a conditional issue, a spill — before the data has arrived; the form with
        /// separate issue and wait statements worked until the register allocation around it changed)
#define LEXLS_LD16 "global_load_dwordx4 "
        template <bool L2LOCAL> __device__ __forceinline__ u32x4 ld16_x(const void *p)
        {
            u32x4 v;
            if (L2LOCAL)
                asm volatile(LEXLS_LD16 "%0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
            else
                asm volatile(LEXLS_LD16 "%0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
            return v;
        }
        template <bool L2LOCAL> __device__ __forceinline__ void ld16_x2(u32x4 &a, u32x4 &b, const void *pa, const void *pb)
        {
            if (L2LOCAL)
                asm volatile(LEXLS_LD16 "%0, %2, off sc1\n\t" LEXLS_LD16 "%1, %3, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(pa), "v"(pb) : "memory");
            else
                asm volatile(LEXLS_LD16 "%0, %2, off sc0 sc1\n\t" LEXLS_LD16 "%1, %3, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(pa), "v"(pb) : "memory");
        }
        template <bool L2LOCAL> __device__ __forceinline__ void ld16_x3(u32x4 &a, u32x4 &b, u32x4 &c, const void *pa, const void *pb, const void *pc)
        {
            if (L2LOCAL)
                asm volatile(LEXLS_LD16 "%0, %3, off sc1\n\t" LEXLS_LD16 "%1, %4, off sc1\n\t" LEXLS_LD16 "%2, %5, off sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(pa), "v"(pb), "v"(pc) : "memory");
            else
                asm volatile(LEXLS_LD16 "%0, %3, off sc0 sc1\n\t" LEXLS_LD16 "%1, %4, off sc0 sc1\n\t" LEXLS_LD16 "%2, %5, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(pa), "v"(pb), "v"(pc) : "memory");
        }
        template <bool L2LOCAL> __device__ __forceinline__ void ld16_x4(u32x4 &a, u32x4 &b, u32x4 &c, u32x4 &d, const void *pa, const void *pb, const void *pc, const void *pd)
        {
            if (L2LOCAL)
                asm volatile(LEXLS_LD16 "%0, %4, off sc1\n\t" LEXLS_LD16 "%1, %5, off sc1\n\t" LEXLS_LD16 "%2, %6, off sc1\n\t" LEXLS_LD16 "%3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(pa), "v"(pb), "v"(pc), "v"(pd) : "memory");
            else
                asm volatile(LEXLS_LD16 "%0, %4, off sc0 sc1\n\t" LEXLS_LD16 "%1, %5, off sc0 sc1\n\t" LEXLS_LD16 "%2, %6, off sc0 sc1\n\t" LEXLS_LD16 "%3, %7, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(pa), "v"(pb), "v"(pc), "v"(pd) : "memory");
        }
        /// Tear-proofing of the 16-byte granules.  An aligned global_store_dwordx4 / global_load_dwordx4 has been observed to travel as one piece
        /// on gfx950 (MI355X_MICROARCH.md, "R2's granule"), but nothing in the ISA text promises it; so each granule's tag half also carries a
        /// checksum of its payload half and of the tag: a reader that saw the new tag with an old payload (or the reverse) finds the checksum
        /// wrong, treats the granule as "not there yet" and asks again (the bounded spins and the abort flag cover the rest).
        /// (as few operations as possible: packing and checking sit on every pivot's critical path; a torn granule pairs halves of two DIFFERENT
        /// publications, so any mixing of the payload's words will do.  Cost on configs[1]: 2.07 -> 2.20 ms; -DLEXLS_PERSIST_NOCHECK builds
        /// without, for A/B.  scripts/persist_stamps.py counts granules whose tag is there and whose checksum is not: none seen so far)
        /// record = {norm (8 bytes) | position (20 bits) + 12 checksum bits | tag (16 bits) + column inside the workgroup (8) + 8 checksum bits}
        constexpr uint32_t kRecNoPos = 0xFFFFFu; // "no candidate"
        __device__ __forceinline__ uint32_t record_check20(uint32_t lo, uint32_t hi, uint32_t p20, uint32_t tag)
        {
            const uint32_t h = lo ^ hi ^ (p20 << 7) ^ tag;
            return (h ^ (h >> 20)) & 0xFFFFFu;
        }
        __device__ __forceinline__ u32x4 record_pack(double norm, uint32_t pos, uint32_t tag, uint32_t idx)
        {
            u32x4 q;
            q.x = (unsigned)__double2loint(norm), q.y = (unsigned)__double2hiint(norm);
            const uint32_t p20 = pos == 0xffffffffu ? kRecNoPos : pos;
            const uint32_t c   = record_check20(q.x, q.y, p20, tag);
            q.z = p20 | ((c >> 8) << 20);
            q.w = (tag << 16) | ((idx & 255u) << 8) | (c & 255u);
            return q;
        }
        __device__ __forceinline__ bool record_ok(const u32x4 &q, uint32_t tag)
        {
#ifdef LEXLS_PERSIST_NOCHECK
            return (q.w >> 16) == tag; // (A/B builds: what the checksums cost)
#endif
            const uint32_t c = record_check20(q.x, q.y, q.z & kRecNoPos, tag);
            return ((q.w >> 16) == tag) & ((q.z >> 20) == (c >> 8)) & ((q.w & 255u) == (c & 255u));
        }
        /// column granule = {value (8 bytes) | tag | low word ^ high word of the value}
        __device__ __forceinline__ u32x4 value_pack(double v, uint32_t tag)
        {
            u32x4 q;
            q.x = (unsigned)__double2loint(v), q.y = (unsigned)__double2hiint(v), q.z = tag;
            q.w = q.x ^ q.y;
            return q;
        }
        __device__ __forceinline__ bool value_ok(const u32x4 &q, uint32_t tag)
        {
#ifdef LEXLS_PERSIST_NOCHECK
            return q.z == tag;
#endif
            return (q.z == tag) & ((q.x ^ q.y) == q.w);
        }

        __device__ __forceinline__ unsigned persist_xcc_id()
        {
            unsigned v;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
            return v & 15u;
        }

        /// NW: wavefronts per workgroup = columns per workgroup (one each).  ONEXCD: the launch is nx times as large as the G workgroups it
        /// needs (nx = XCDs of the device; workgroups go to the XCDs round robin); the workgroups that find themselves on XCD 0 claim the tile
        /// indices in arrival order, all others leave at once — so every hand-off stays inside one L2 (see st16_x)
        template <int NW, int CPW, bool ONEXCD>
        __global__ __launch_bounds__(64 * NW) void fast_level_persist(LseArgs a, FastBuffers fb, PersistCtl *ctl, PersistCand *cand, double *colbuf, uint32_t colld,
                                                                      uint32_t cur, uint32_t pp, uint32_t level, uint32_t G)
        {
            constexpr int PTC      = NW * CPW; // columns per workgroup, CPW per wavefront
            constexpr uint32_t NT  = 64u * NW;
            // Records travel by MAILBOX: a workgroup stores its record once per reader, into that reader's own row of G consecutive slots
            // (lane = reader: three store instructions of wave 0), and a reader polls only its own row — 2 KB that nobody else reads, as
            // coalesced loads.  Before (every workgroup's ONE record polled by all G: 129 x 129 scattered 16-byte reads of 129 hot lines per
            // round) a poll round trip took 2200 cycles against 900 for a single line (scripts/ubench/xcd_pingpong.hip).
            const size_t MBS = persist_mailbox_stride(G);
            extern __shared__ double smem[];
            __shared__ u32x4 slot[NW]; // the wavefronts' own candidates of the next pivot: {norm, position, column}
            __shared__ uint32_t flag;
            __shared__ uint32_t win_p, win_i, win_w, colbad, claimed;
            const uint32_t b = 0, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
            const LargeState s = fb.st[pp][b];
            const uint32_t n = a.nVar, cap = a.cap;
            if (s.exhausted || s.dim == 0) return; // the same in every workgroup: nobody waits for anybody
            if (__hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return; // (raised before the launch: the tests' way into the fallback)
            uint32_t t = blockIdx.x;
            if (ONEXCD)
            {
                if (persist_xcc_id() != 0u) return;
                if (tid == 0) claimed = atomicAdd(&ctl->arrive, 1u);
                __syncthreads();
                t = claimed;
                if (t >= G) return;
            }
            const uint32_t dim = s.dim, F = s.F;
            double *W          = fb.W[cur] + (size_t)b * cap * (n + 1);
            const uint32_t ldt = dim | 1u;
            double *tile   = smem;                      // PTC x ldt
            double *cv0    = tile + (size_t)PTC * ldt;  // 2 x dim: the pivot column, by pivot parity (the previous pivot's essential part leaves one step late)
            uint32_t *posm = reinterpret_cast<uint32_t *>(cv0 + 2 * dim); // n + 1: position of every physical column (kept by every workgroup)
            uint32_t *invm = posm + (n + 1);                              // n + 1: physical column at every position

            for (uint32_t e = tid; e < (uint32_t)PTC * dim; e += NT)
            {
                const uint32_t jj = e / dim, i = e - jj * dim, j = t * PTC + jj;
                tile[jj * ldt + i] = (j <= n) ? W[F + i + (size_t)j * cap] : 0.0;
            }
            for (uint32_t j = tid; j <= n; j += NT)
            {
                posm[j] = j;
                invm[j] = j;
            }
            if (tid == 0) colbad = 0u;
            // The wavefront's own columns: physical index, position and down-dated norm live in (wave-uniform) registers — what the next pivot's
            // record depends on never goes through LDS except for the one exchange between the wavefronts of the workgroup (slot[])
            uint32_t jq[CPW], mypos[CPW];
            double mynrm[CPW];
#pragma unroll
            for (int q = 0; q < CPW; q++)
            {
                jq[q]    = t * PTC + wave * CPW + q;
                mypos[q] = jq[q];
                mynrm[q] = (jq[q] < n) ? fb.norms[pp][(size_t)b * n + jq[q]] : -1.0;
            }
            __syncthreads();

            uint32_t c = s.ColIndex, rank = 0, stop = 0, exhausted = 0;
#ifdef LEXLS_PERSIST_STAMPS
            long long pst[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pt0 = clock64();
#define PSTAMP(i) { const long long t_ = clock64(); pst[i] += t_ - pt0; pt0 = t_; }
#else
#define PSTAMP(i)
#endif
            // bookkeeping of a pivot (its essential part, R_cc, permutation entry, tau) is written by ONE workgroup, a different one every
            // pivot, and ONE STEP LATE: behind that workgroup's next publication instead of in front of it, where its stores would make it
            // the last to publish — and everybody waits for the last.  The essential part is formed then, from the kept pivot column.
            bool pend = false, pend_degenerate = false;
            uint32_t pend_counter = 0, pend_c = 0, pend_ppos = 0, pend_R = 0;
            double pend_diag = 0.0, pend_tau = 0.0, pend_iden = 1.0;
            auto flush_pending = [&]() {
                if (!pend) return;
                const double *cvp = cv0 + (size_t)(pend_counter & 1u) * dim;
                double *E         = fb.E + ((size_t)b * fb.eld + pend_counter) * fb.eld;
                for (uint32_t i = 1 + tid; i < pend_R; i += NT) E[i] = pend_degenerate ? 0.0 : cvp[i] * pend_iden;
                if (tid == 0)
                {
                    fb.D[(size_t)b * n + pend_c]   = pend_diag;
                    a.perm[(size_t)b * n + pend_c] = pend_ppos;
                    if (pend_R > 1) a.hh[(size_t)b * cap + F + pend_counter] = pend_tau;
                }
                pend = false;
            };
            // Own candidate of pivot `cnt` (first maximum by position among the workgroup's live columns) and its RECORD — published as early as
            // the norms allow, before the tile update of the previous pivot has finished: the others wait for records, the candidate's column
            // follows (publish_column) and is only read by those who need it.  Every 16-byte store carries the pivot's tag: nothing is drained and
            // nothing is ordered — a reader takes a granule when its tag matches.  The wavefronts exchange their own best through slot[] (one
            // barrier); a scan of the workgroup's columns through LDS by every thread cost 370 cycles per column on this critical path.
            uint32_t cand_p = 0xffffffffu, cand_j = 0;
            auto publish_record = [&](uint32_t cnt, uint32_t cfirst) {
                // a candidate = {squared norm, key}, key = position << 8 | column inside the workgroup (positions are distinct: the smallest key is
                // the smallest position), 0xffffffff = none; "first maximum by position" = largest norm, then smallest key — as a maximum
                // tree and a minimum tree (a compare-and-select chain over the wavefronts cost ~150 cycles per wavefront)
                double wv   = -1.0; // (norms are >= 0: -1 = no candidate)
                uint32_t wk = 0xffffffffu;
#pragma unroll
                for (int q = 0; q < CPW; q++)
                {
                    const uint32_t kq = (mypos[q] << 8) | (uint32_t)(wave * CPW + q);
                    const bool gt = (jq[q] < n) & (mypos[q] >= cfirst) & ((mynrm[q] > wv) | ((mynrm[q] == wv) & (kq < wk))); // (no short-circuit branches)
                    wv = gt ? mynrm[q] : wv, wk = gt ? kq : wk;
                }
                if (lane == 0)
                {
                    u32x4 sq;
                    sq.x = (unsigned)__double2loint(wv), sq.y = (unsigned)__double2hiint(wv), sq.z = wk, sq.w = 0u;
                    slot[wave] = sq;
                }
                __syncthreads(); // C: the wavefronts' candidates (and the maps of this pivot) are in place
                PSTAMP(8)
                u32x4 sv[NW];
                double vv[NW];
#pragma unroll
                for (int w2 = 0; w2 < NW; w2++) sv[w2] = slot[w2];
#pragma unroll
                for (int w2 = 0; w2 < NW; w2++) vv[w2] = __hiloint2double((int)sv[w2].y, (int)sv[w2].x);
                double mx[NW];
#pragma unroll
                for (int w2 = 0; w2 < NW; w2++) mx[w2] = vv[w2];
#pragma unroll
                for (int h = NW / 2; h >= 1; h /= 2)
#pragma unroll
                    for (int w2 = 0; w2 < h; w2++) mx[w2] = wave_vmax(mx[2 * w2], mx[2 * w2 + 1]);
                const double myv = mx[0];
                uint32_t kk[NW];
#pragma unroll
                for (int w2 = 0; w2 < NW; w2++) kk[w2] = vv[w2] == myv ? sv[w2].z : 0xffffffffu;
#pragma unroll
                for (int h = NW / 2; h >= 1; h /= 2)
#pragma unroll
                    for (int w2 = 0; w2 < h; w2++) kk[w2] = kk[2 * w2] < kk[2 * w2 + 1] ? kk[2 * w2] : kk[2 * w2 + 1];
                const uint32_t myk = kk[0];
                const uint32_t myp = myk == 0xffffffffu ? 0xffffffffu : (myk >> 8), myj = t * PTC + (myk & 255u);
                cand_p = myp, cand_j = myj;
                if (wave == 0)
                {
                    const u32x4 q = record_pack(myv, myp, cnt + 1u, myk & 255u); // (no candidate: the position says so, the index is not looked at)
                    PersistCand *box = cand + (size_t)(cnt & 1u) * G * MBS + t; // slot t of every reader's row
#pragma unroll
                    for (int kq = 0; kq < 4; kq++)
                    {
                        const uint32_t reader = lane + 64u * kq;
                        if (reader < G) st16_x<ONEXCD>(box + (size_t)reader * MBS, q);
                    }
                }
            };
            // the candidate's remaining rows, by the wavefront that owns (and has just updated) the column — and, behind them, the column's
            // fresh squared norm and the squared norm of its tail (rows 1..): every reader needs both for the reflector, the owner forms
            // them while nobody waits for it
            auto publish_column = [&](uint32_t cnt) {
                if (cand_p == 0xffffffffu || (cand_j - t * PTC) / CPW != wave) return;
                const uint32_t Rn  = dim - cnt;
                u32x4 *mycol       = reinterpret_cast<u32x4 *>(colbuf) + ((size_t)(cnt & 1u) * G + t) * colld;
                const double *srcc = tile + (cand_j - t * PTC) * ldt + cnt;
                double fr = 0.0, tl = 0.0;
                for (uint32_t i = lane; i < Rn; i += 64)
                {
                    const double v0 = srcc[i];
                    st16_x<ONEXCD>(mycol + i, value_pack(v0, cnt + 1u));
                    fr = dfma(v0, v0, fr);
                    if (i > 0) tl = dfma(v0, v0, tl);
                }
                fr = wave_sum(fr);
                tl = wave_sum(tl);
                if (lane < 2)
                {
                    const double v0 = lane == 0 ? fr : tl;
                    st16_x<ONEXCD>(mycol + Rn + lane, value_pack(v0, cnt + 1u));
                }
            };
            publish_record(0, c);
            publish_column(0);
            for (uint32_t counter = 0; counter < dim; counter++)
            {
                const uint32_t R = dim - counter, par = counter & 1u;
                double *colv     = cv0 + (size_t)par * dim;
                const uint32_t tag = counter + 1u;
                flush_pending(); // (the previous pivot's bookkeeping, if it was this workgroup's turn: nobody waits for these stores)
                const uint32_t front = invm[c]; // the column at the pivot's position (written before the last barrier C)
                PSTAMP(0)
                // ---- wait for the G records of this pivot and pick the winner: wave 0, lane = workgroup (no counter: a record IS its flag) ----
                if (wave == 0)
                {
                    // every lane keeps up to four records (G <= 256) in flight per poll: ONE round trip per poll, not one per record
                    const PersistCand *base = cand + ((size_t)par * G + t) * MBS; // this workgroup's row
                    u32x4 q[4] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};
                    uint32_t ok = 0;
                    for (uint32_t spin = 0; spin < (1u << 18); spin++)
                    {
                        {
                            const PersistCand *a0 = base + lane, *a1 = base + (lane + 64u < G ? lane + 64u : 0u), *a2 = base + (lane + 128u < G ? lane + 128u : 0u),
                                              *a3 = base + (lane + 192u < G ? lane + 192u : 0u);
                            if (G <= 64u) // (wave-uniform: as many loads as the row has records)
                                q[0] = ld16_x<ONEXCD>(base + (lane < G ? lane : 0u));
                            else if (G <= 128u)
                                ld16_x2<ONEXCD>(q[0], q[1], a0, a1);
                            else if (G <= 192u)
                                ld16_x3<ONEXCD>(q[0], q[1], q[2], a0, a1, a2);
                            else
                                ld16_x4<ONEXCD>(q[0], q[1], q[2], q[3], a0, a1, a2, a3);
                        }
                        bool mine = true;
#pragma unroll
                        for (int kq = 0; kq < 4; kq++) mine = mine && (lane + 64u * kq >= G || (q[kq].w >> 16) == tag);
                        if (__ballot(!mine) == 0ull) // every tag is there: now the checksums (once per pivot, not once per poll)
                        {
#pragma unroll
                            for (int kq = 0; kq < 4; kq++) mine = mine && (lane + 64u * kq >= G || record_ok(q[kq], tag));
                        }
#ifdef LEXLS_PERSIST_STAMPS
                        pst[2] += 1; // polls
#ifdef LEXLS_PERSIST_TORN_COUNT
                        {
                            bool torn = false; // the tag is there, the checksum is not: halves of two publications
#pragma unroll
                            for (int kq = 0; kq < 4; kq++) torn = torn || (lane + 64u * kq < G && (q[kq].w >> 16) == tag && !record_ok(q[kq], tag));
                            if (__ballot(torn) != 0ull) pst[12] += 1;
                        }
#endif
#endif
                        if (__ballot(!mine) == 0ull)
                        {
                            ok = 1;
                            break;
                        }
                        if ((spin & 255u) == 255u && __hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                    double bv   = -1.0;
                    uint32_t bp = 0xffffffffu, bi = 0, bw = 0;
#pragma unroll
                    for (int kq = 0; kq < 4; kq++)
                    {
                        const uint32_t w = lane + 64u * kq;
                        const double v   = __hiloint2double((int)q[kq].y, (int)q[kq].x);
                        const uint32_t p = (q[kq].z & kRecNoPos) == kRecNoPos ? 0xffffffffu : (q[kq].z & kRecNoPos);
                        if (ok && w < G && p != 0xffffffffu && (v > bv || (v == bv && p < bp)))
                        {
                            bv = v;
                            bp = p;
                            bi = w * PTC + ((q[kq].w >> 8) & 255u);
                            bw = w;
                        }
                    }
                    const bool all_ok = ok != 0;
                    if (!all_ok && lane == 0) __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const double wm         = wave_max(bv);
                    unsigned long long tied = __ballot(bv == wm && bp != 0xffffffffu);
                    uint32_t bestp = 0xffffffffu, besti = 0, bestw = 0;
                    while (tied)
                    {
                        const int l = (int)__builtin_ctzll(tied);
                        tied &= tied - 1;
                        const uint32_t p2 = (uint32_t)__builtin_amdgcn_readlane((int)bp, l);
                        if (p2 < bestp)
                        {
                            bestp = p2;
                            besti = (uint32_t)__builtin_amdgcn_readlane((int)bi, l);
                            bestw = (uint32_t)__builtin_amdgcn_readlane((int)bw, l);
                        }
                    }
                    if (lane == 0)
                    {
                        flag  = all_ok ? 1u : 0u;
                        win_p = bestp;
                        win_i = besti;
                        win_w = bestw;
                    }
                }
                __syncthreads(); // A
                if (!flag) return; // (uniform per workgroup; the others see `abort`)
                const uint32_t ppos = win_p, piv = win_i, wwin = win_w;
                PSTAMP(1)

                // ---- the pivot column (published by its owner) into LDS; every thread also takes the two sums behind it ----
                const u32x4 *pcol = reinterpret_cast<const u32x4 *>(colbuf) + ((size_t)par * G + wwin) * colld;
                // this thread's rows of the column and the two sums behind it: all requests in flight together, ONE wait; a granule whose tag
                // does not match yet is asked for again (the record may overtake its column)
                constexpr uint32_t CR = 1; // rows per thread in the batch (levels of more than 64 NW rows: the rest one by one, below)
                double fresh = 0.0, tailSq = 0.0;
                {
                    u32x4 g[CR + 2];
                    uint32_t good = 0;
                    for (uint32_t spin = 0; spin < (1u << 18); spin++)
                    {
                        static_assert(CR == 1, "one row of the column per thread in the batch");
                        ld16_x3<ONEXCD>(g[0], g[1], g[2], pcol + (tid < R ? tid : R), pcol + R, pcol + R + 1u);
                        bool all = true;
#pragma unroll
                        for (uint32_t r = 0; r < CR + 2; r++) all = all & value_ok(g[r], tag);
#ifdef LEXLS_PERSIST_STAMPS
                        pst[11] += 1; // column read rounds
#ifdef LEXLS_PERSIST_TORN_COUNT
                        if ((g[0].z == tag && !value_ok(g[0], tag)) || (g[1].z == tag && !value_ok(g[1], tag)) || (g[2].z == tag && !value_ok(g[2], tag))) pst[13] += 1; // torn value granules seen by this thread
#endif
#endif
                        if (all)
                        {
                            good = 1;
                            break;
                        }
                        if ((spin & 255u) == 255u && __hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                    }
                    if (!good)
                    {
                        __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        colbad = 1u;
                    }
#pragma unroll
                    for (uint32_t r = 0; r < CR; r++)
                    {
                        const uint32_t i = tid + r * NT;
                        if (i < R) colv[i] = __hiloint2double((int)g[r].y, (int)g[r].x);
                    }
                    fresh  = __hiloint2double((int)g[CR].y, (int)g[CR].x);
                    tailSq = __hiloint2double((int)g[CR + 1].y, (int)g[CR + 1].x);
                }
                for (uint32_t i = tid + CR * NT; i < R; i += NT)
                {
                    u32x4 q;
                    uint32_t good = 0;
                    for (uint32_t spin = 0; spin < (1u << 18); spin++)
                    {
                        q = ld16_x<ONEXCD>(pcol + i);
                        if (value_ok(q, tag))
                        {
                            good = 1;
                            break;
                        }
                        if ((spin & 255u) == 255u && __hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                    }
                    if (!good)
                    {
                        __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        colbad = 1u;
                    }
                    colv[i] = __hiloint2double((int)q.y, (int)q.x);
                }
                __syncthreads(); // B
                PSTAMP(3)
                if (colbad) return; // (a column granule timed out in this workgroup; `abort` is raised, the others see it in their spins)

                // ---- own tile, first half: the dot product of the wave's columns with the RAW pivot column — it does not wait for the
                //      reflector's scalars, it runs beside them.  (This path's contract is exact pivots and values within 1e-10:
                //      es^T a = (v^T a) / den is the same number up to rounding.) ----
                // the column swap of this pivot (below) gives `piv` position c and the column that was there position ppos: trailing afterwards
                // is "not the pivot and not finished before"
                double *col[CPW];
                bool trailing[CPW];
                double part[CPW], dotraw[CPW], a0[CPW];
#pragma unroll
                for (int q = 0; q < CPW; q++)
                {
                    col[q]      = tile + (wave * CPW + q) * ldt + counter;
                    trailing[q] = jq[q] == n || (jq[q] < n && jq[q] != piv && mypos[q] >= c);
                    part[q]     = 0.0;
                    a0[q]       = col[q][0];
                }
                for (uint32_t i = 1 + lane; i < R; i += 64)
                {
                    const double cvi = colv[i];
#pragma unroll
                    for (int q = 0; q < CPW; q++) part[q] = dfma(cvi, col[q][i], part[q]);
                }
#pragma unroll
                for (int q = 0; q < CPW; q++) dotraw[q] = wave_sum(part[q]); // (columns that are not trailing: the value is not used)
                PSTAMP(5)

                if (fresh < a.tol) // rank test (lexlse.h:214): the level ends here, in every workgroup
                {
                    stop = 1;
                    break;
                }
                const double c0v = colv[0];
                // The reflector's scalars WITHOUT the division / square-root sequences (three of them, ~400 cycles each, sat on every pivot's
                // critical path): |beta| = sqrt(nn) and 1 / |beta| from v_rsq_f64 + two coupled Newton steps, 1 / den from v_rcp_f64 + two
                // steps — within an ulp or two of the correctly rounded values, which this path's contract (values within 1e-10) allows
                double tau = 0.0, diag = c0v, iden = 1.0;
                bool degenerate = true;
                if (R > 1 && !(tailSq <= DBL_MIN))
                {
                    degenerate      = false;
                    const double nn = dfma(c0v, c0v, tailSq);
                    double y        = __builtin_amdgcn_rsq(nn);
                    double g = nn * y, h = 0.5 * y;
                    double r = dfma(-h, g, 0.5);
                    g        = dfma(g, r, g);
                    h        = dfma(h, r, h);
                    r        = dfma(-h, g, 0.5);
                    h        = dfma(h, r, h);
                    g        = dfma(dfma(-g, g, nn), h, g); // |beta|;  2 h = 1 / |beta|
                    const double beta = c0v >= 0.0 ? -g : g;
                    diag              = beta;
                    tau               = dfma(fabs(c0v), 2.0 * h, 1.0); // (beta - c0v) / beta
                    const double den  = c0v - beta;
                    double ri         = __builtin_amdgcn_rcp(den);
                    ri                = dfma(dfma(-den, ri, 1.0), ri, ri);
                    iden              = dfma(dfma(-den, ri, 1.0), ri, ri);
                }
                PSTAMP(6)
                // the swap of lexlse.h:222-232 on both maps (every workgroup keeps them; the wavefronts keep their own columns' positions)
                if (tid == 0)
                {
                    posm[piv]   = c;
                    invm[c]     = piv;
                    if (front != piv)
                    {
                        posm[front] = ppos;
                        invm[ppos]  = front;
                    }
                }
                // ---- own tile, second half: the new pivot-row entry and the down-dated norm first — they decide the NEXT pivot's record ----
                double sc[CPW];
                bool upd[CPW];
#pragma unroll
                for (int q = 0; q < CPW; q++)
                {
                    upd[q] = trailing[q] && tau != 0.0;
                    sc[q]  = 0.0;
                    if (trailing[q])
                    {
                        double a0n = a0[q];
                        if (tau != 0.0)
                        {
                            const double tmp = dfma(dotraw[q], iden, a0[q]);
                            a0n              = dfma(-tau, tmp, a0[q]);
                            sc[q]            = (-tau * tmp) * iden;
                            if (lane == 0) col[q][0] = a0n;
                        }
                        if (jq[q] < n) mynrm[q] = dfma(-a0n, a0n, mynrm[q]);
                    }
                    if (jq[q] == piv)
                        mypos[q] = c;
                    else if (jq[q] == front)
                        mypos[q] = ppos;
                }
                if (t == counter % G)
                {
                    pend         = true;
                    pend_counter = counter, pend_c = c, pend_ppos = ppos, pend_R = R;
                    pend_diag = diag, pend_tau = tau, pend_iden = iden;
                    pend_degenerate = degenerate;
                }
                c++;
                rank++;
                PSTAMP(7)
                const bool last = c == n || counter + 1 == dim;
                if (!last)
                    publish_record(counter + 1, c);
                else
                    __syncthreads(); // (the barrier of publish_record: the maps are complete before they are written back)
                PSTAMP(9)
                // ---- the rest of the columns, then (if one of them is the candidate) its rows for the others ----
                for (uint32_t i = 1 + lane; i < R; i += 64)
                {
                    const double cvi = colv[i];
#pragma unroll
                    for (int q = 0; q < CPW; q++)
                        if (upd[q]) col[q][i] = dfma(sc[q], cvi, col[q][i]);
                }
                PSTAMP(10)
                if (!last) publish_column(counter + 1);
                PSTAMP(4)
                if (c == n)
                {
                    exhausted = 1;
                    break;
                }
            }
#ifdef LEXLS_PERSIST_STAMPS
            if (t == 1 && tid == 0)
                for (int i_ = 0; i_ < 14; i_++) a.lambda[16 * level + i_] = (double)pst[i_];
            if (tid == 0 && level == 0 && 64u + 2u * G < a.cap) // where the workgroups ran and how long each waited for the records
            {
                unsigned hw;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                a.lambda[64 + 2 * t]     = (double)hw;
                a.lambda[64 + 2 * t + 1] = (double)pst[1];
            }
#endif
            // ---- back to memory: the tile, the position map, the state — unless some workgroup gave up (then nobody commits: a workgroup that
            //      timed out did so long before any other could finish the level, every later hand-off needs its record) ----
            // TWO-PHASE: the tile is committed IN PLACE, so "nobody commits if anyone gave up" must not rest on timing — a
            // workgroup could run out of spins on the LAST pivot's hand-off after others had sampled `abort` and written their tiles, and the
            // fall-back would start from a partly transformed level.  Every workgroup that got through the level arrives at `done`; a tile is
            // written only once all G have arrived (nobody is left who could still give up) and `abort` is still clear.
            __syncthreads();
            if (tid == 0)
            {
                uint32_t ok = 0u;
                if (!__hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                {
                    __hip_atomic_fetch_add(&ctl->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (uint32_t spin = 0; spin < (1u << 20); spin++)
                    {
                        if (__hip_atomic_load(&ctl->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= G)
                        {
                            ok = 1u;
                            break;
                        }
                        if ((spin & 63u) == 63u && __hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (!ok) __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (the others' spins end; a late arriver finds it)
                    else if (__hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ok = 0u;
                }
                flag = ok;
            }
            __syncthreads();
            if (!flag) return;
            flush_pending();
            for (uint32_t e = tid; e < (uint32_t)PTC * dim; e += NT)
            {
                const uint32_t jj = e / dim, i = e - jj * dim, j = t * PTC + jj;
                if (j <= n) W[F + i + (size_t)j * cap] = tile[jj * ldt + i];
            }
            if (t == 0)
            {
                uint32_t *pos_out = fb.pos[pp ^ 1u] + (size_t)b * (n + 1);
                for (uint32_t j = tid; j <= n; j += NT) pos_out[j] = posm[j];
                if (tid == 0)
                {
                    LargeState so   = s;
                    so.ColIndex     = c;
                    so.rank         = rank;
                    so.exhausted    = exhausted;
                    so.stop_level   = stop;
                    so.last_id      = pp ^ 1u;
                    fb.st[pp ^ 1u][b] = so;
                }
            }
        }

        /// level end: rank / first column, then the whole matrix in the level's final column order -> the other work buffer
        __global__ __launch_bounds__(256) void fast_level_end(LseArgs a, FastBuffers fb, uint32_t cur, uint32_t pp, uint32_t level)
        {
            const uint32_t b = blockIdx.z;
            if (skipped(a, b)) return;
            const uint32_t n = a.nVar, cap = a.cap, nObj = a.nObj;
            const uint32_t *dims = a.dims + (size_t)b * nObj;
            uint32_t M           = 0;
            for (uint32_t k = 0; k < nObj; k++) M += dims[k];
            const LargeState *s = fb.st[pp] + b;
            const uint32_t F = s->F, dim = s->dim, Fc = s->Fc, rank = s->rank;
            const uint32_t j = blockIdx.y; // physical column
            const uint32_t p = (j == n) ? n : fb.pos[s->last_id & 1u][(size_t)b * (n + 1) + j];
            const double *src = fb.W[cur] + (size_t)b * cap * (n + 1) + (size_t)j * cap;
            double *dst       = fb.W[cur ^ 1u] + (size_t)b * cap * (n + 1) + (size_t)p * cap;
            const bool pivcol = j < n && p >= Fc && p < Fc + rank;
            const uint32_t q  = p - Fc; // pivot index inside the level (pivot q was made at step q)
            const double *E   = fb.E + ((size_t)b * fb.eld + (pivcol ? q : 0)) * fb.eld;
            for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < M; i += gridDim.x * 256)
            {
                double v = src[i];
                if (pivcol && i >= F + q && i < F + dim)
                    v = (i == F + q) ? fb.D[(size_t)b * n + p] : E[i - (F + q)];
                dst[i] = v;
            }
            if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
            {
                a.rank[(size_t)b * nObj + level] = rank;
                a.fcol[(size_t)b * nObj + level] = Fc;
            }
        }

        __global__ void fast_level_commit(LseArgs a, FastBuffers fb, uint32_t pp)
        {
            const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
            if (b >= a.batch || skipped(a, b)) return;
            LargeState *s = fb.st[pp] + b;
            s->totalrank += s->rank;
            s->stop_level  = 0;
            a.totalrank[b] = s->totalrank;
        }

        __global__ __launch_bounds__(256) void fast_copy_back(LseArgs a, FastBuffers fb, uint32_t cur)
        {
            const uint32_t b = blockIdx.y;
            if (skipped(a, b)) return;
            const size_t ps   = (size_t)a.cap * (a.nVar + 1);
            const double *src = fb.W[cur] + b * ps;
            double *dst       = fb.W[0] + b * ps;
            for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < ps; e += (size_t)gridDim.x * 256) dst[e] = src[e];
        }

        /// Trailing -= L * Up (lexlse.h:454-469) on the matrix cores: 64 x 64 output block per workgroup, K in steps of 16 through LDS.
        /// acc = C, then for p ascending acc = fma(-L[i][p], U[p][j], acc): v_mfma_f64_16x16x4_f64 does exactly that for four p at a time.
        constexpr int GBM = 64, GBN = 64, GBK = 16;
        __global__ __launch_bounds__(256) void large_gemm_mfma(LseArgs a, const LargeState *st, uint32_t level)
        {
            __shared__ double As[GBK * GBM]; // As[k][i] = -L[i][k]
            __shared__ double Bs[GBK * GBN]; // Bs[k][j] =  U[k][j]
            const uint32_t b = blockIdx.z, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
            if (skipped(a, b)) return;
            const LargeState *s = st + b;
            const uint32_t rank = s->rank;
            if (rank == 0 || level + 1 >= a.nObj) return;
            const uint32_t n = a.nVar, cap = a.cap, nObj = a.nObj;
            const uint32_t *dims = a.dims + (size_t)b * nObj;
            uint32_t M           = 0;
            for (uint32_t k = 0; k < nObj; k++) M += dims[k];
            const uint32_t F = s->F, Fc = s->Fc, Fn = F + s->dim, c = s->ColIndex;
            const uint32_t gi0 = Fn + blockIdx.x * GBM; // first row of the block
            const uint32_t j0  = c + blockIdx.y * GBN;  // first column of the block
            if (gi0 >= M || j0 > n) return;            // uniform per workgroup
            double *W = a.fac + (size_t)b * cap * (n + 1);

            // accumulators: tile t = columns j0 + 16 t .. +15, rows gi0 + 16 wave .. +15; lane l, entry r: row (l / 16) + 4 r, column l % 16
            const uint32_t ri = gi0 + 16 * wave + (lane >> 4);
            v4f64 acc[4];
#pragma unroll
            for (int t = 0; t < 4; t++)
            {
                const uint32_t j = j0 + 16 * t + (lane & 15u);
#pragma unroll
                for (int r = 0; r < 4; r++) acc[t][r] = (j <= n && ri + 4 * r < M) ? W[ri + 4 * r + (size_t)j * cap] : 0.0;
            }
            const uint32_t K4 = rank & ~3u;
            for (uint32_t k0 = 0; k0 < K4; k0 += GBK)
            {
                __syncthreads();
                // stage: 256 threads, 4 elements each per operand (coalesced along the rows of W)
#pragma unroll
                for (int u = 0; u < 4; u++)
                {
                    const uint32_t i = tid & 63u, k = (tid >> 6) + 4 * u; // A: 64 rows x 16 k
                    As[k * GBM + i]  = (gi0 + i < M && k0 + k < K4) ? -W[gi0 + i + (size_t)(Fc + k0 + k) * cap] : 0.0;
                    const uint32_t kb = tid & 15u, jb = (tid >> 4) + 16 * u; // B: 16 k x 64 columns
                    Bs[kb * GBN + jb] = (j0 + jb <= n && k0 + kb < K4) ? W[F + k0 + kb + (size_t)(j0 + jb) * cap] : 0.0;
                }
                __syncthreads();
                const uint32_t ksteps = (K4 - k0 < (uint32_t)GBK ? K4 - k0 : (uint32_t)GBK) / 4;
                for (uint32_t ks = 0; ks < ksteps; ks++)
                {
                    const double af = As[(4 * ks + (lane >> 4)) * GBM + 16 * wave + (lane & 15u)];
#pragma unroll
                    for (int t = 0; t < 4; t++)
                    {
                        const double bf = Bs[(4 * ks + (lane >> 4)) * GBN + 16 * t + (lane & 15u)];
                        acc[t]          = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc[t], 0, 0, 0);
                    }
                }
            }
            // the last rank % 4 pivots: plain fma's on the accumulator layout
            for (uint32_t p = K4; p < rank; p++)
            {
#pragma unroll
                for (int t = 0; t < 4; t++)
                {
                    const uint32_t j = j0 + 16 * t + (lane & 15u);
                    const double u   = (j <= n) ? W[F + p + (size_t)j * cap] : 0.0;
#pragma unroll
                    for (int r = 0; r < 4; r++)
                    {
                        const double l = (ri + 4 * r < M) ? W[ri + 4 * r + (size_t)(Fc + p) * cap] : 0.0;
                        acc[t][r]      = dfma(-l, u, acc[t][r]);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 4; t++)
            {
                const uint32_t j = j0 + 16 * t + (lane & 15u);
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (j <= n && ri + 4 * r < M) W[ri + 4 * r + (size_t)j * cap] = acc[t][r];
            }
        }
    } // namespace

    namespace
    {
        /// dynamic LDS of the three kernels that stage in LDS, for the largest level dimension of the batch (ONE formula for the
        /// dispatcher's question "does it fit" and for the launch)
        struct LargeLds
        {
            size_t piv, app, trsm;
        };
        inline LargeLds large_lds_bytes(uint32_t n, uint32_t maxdim)
        {
            LargeLds l;
            l.piv  = 8 * ((size_t)((maxdim + 1) & ~1u) + 1024 + 16) + 4 * 1024;
            l.app  = 8 * ((size_t)TC * (maxdim | 1u) + maxdim + TC + 2);
            l.trsm = 8 * (size_t)((n < maxdim) ? n : maxdim) * 65;
            return l;
        }
    } // namespace

    bool large_kernel_supports(const LseArgs &a, uint32_t max_level_dim, bool has_fixed)
    {
        const LargeLds l = large_lds_bytes(a.nVar, max_level_dim);
        return !has_fixed && l.trsm <= kMaxLdsBytes && l.piv <= kMaxLdsBytes && l.app <= kMaxLdsBytes && max_level_dim < 65536 && a.nObj < 65536;
    }

    size_t large_state_bytes(uint32_t batch) { return sizeof(LargeState) * (size_t)batch; }

    /// h_level_max[k] = max over the batch of dims[k]; h_rows_max = max rows of one problem
    hipError_t launch_lqr_large(const LseArgs &a, const uint32_t *h_level_max, uint32_t h_rows_max, void *d_state, double *d_norms, hipStream_t s)
    {
        LargeState *st    = static_cast<LargeState *>(d_state);
        const uint32_t B  = a.batch, n = a.nVar;
        hipError_t e      = hipSuccess;
        auto set_lds      = [&](const void *k, size_t bytes) {
            if (e == hipSuccess && bytes > 64 * 1024) e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        };
        uint32_t maxdim = 0;
        for (uint32_t k = 0; k < a.nObj; k++) maxdim = h_level_max[k] > maxdim ? h_level_max[k] : maxdim;
        const LargeLds lds    = large_lds_bytes(n, maxdim);
        const size_t piv_lds = lds.piv, app_lds = lds.app, trsm_lds = lds.trsm;
        set_lds(reinterpret_cast<const void *>(large_pivot), piv_lds);
        set_lds(reinterpret_cast<const void *>(large_apply), app_lds);
        set_lds(reinterpret_cast<const void *>(large_trsm), trsm_lds);
        if (e != hipSuccess) return e;

        hipLaunchKernelGGL(large_init, dim3(64, B), dim3(256), 0, s, a, st);
        std::vector<LargeState> host(B);
        bool all_exhausted = false;
        for (uint32_t level = 0; level < a.nObj; level++)
        {
            hipLaunchKernelGGL(large_level_begin, dim3((n + 63) / 64, B), dim3(64), 0, s, a, st, d_norms, level);
            if (!all_exhausted)
                for (uint32_t counter = 0; counter < h_level_max[level]; counter++)
                {
                    hipLaunchKernelGGL(large_pivot, dim3(1, B), dim3(NTP), piv_lds, s, a, st, d_norms, level, counter);
                    hipLaunchKernelGGL(large_apply, dim3((n + TC) / TC, B), dim3(256), app_lds, s, a, st, d_norms, level, counter);
                }
            hipLaunchKernelGGL(large_level_end, dim3((B + 63) / 64), dim3(64), 0, s, a, st, level);
            // rows below the level: a ragged batch may hold a problem with small early levels and many rows below — the grid spans the
            // largest row count of the batch, workgroups beyond a problem's own rows return at once (r0 >= M)
            const uint32_t below = h_rows_max;
            if (level + 1 < a.nObj && below > 0 && h_level_max[level] > 0) // (a level that is empty in every problem has rank 0: no Gauss step)
            {
                if (h_level_max[level] <= 1024)
                    hipLaunchKernelGGL(large_trsm_cols, dim3((below + TRB - 1) / TRB, B), dim3(((h_level_max[level] + 63) / 64) * 64), 0, s, a, st, level);
                else
                    hipLaunchKernelGGL(large_trsm, dim3((below + 63) / 64, B), dim3(64), trsm_lds, s, a, st, level);
                hipLaunchKernelGGL(large_gemm_mfma, dim3((below + GBM - 1) / GBM, (n + GBN) / GBN, B), dim3(256), 0, s, a, st, level); // bit-identical to large_gemm
            }
            e = hipGetLastError();
            if (e != hipSuccess) return e;
            if (!all_exhausted && level + 1 < a.nObj) // one small read-back per level: stop launching pivots once no column is left anywhere
            {
                e = hipMemcpyAsync(host.data(), st, sizeof(LargeState) * B, hipMemcpyDeviceToHost, s);
                if (e != hipSuccess) return e;
                e = hipStreamSynchronize(s);
                if (e != hipSuccess) return e;
                all_exhausted = (a.skip == nullptr); // skipped problems carry stale state: never stop early then
                for (uint32_t b = 0; b < B && all_exhausted; b++)
                    if (!host[b].exhausted) all_exhausted = false;
            }
        }
        hipLaunchKernelGGL(large_finish, dim3((B + 63) / 64), dim3(64), 0, s, a, st);
        return hipGetLastError();
    }
    namespace
    {
        __global__ __launch_bounds__(64) void persist_probe_xcc(uint32_t *out)
        {
            if (threadIdx.x == 0) out[blockIdx.x] = persist_xcc_id();
        }
        /// XCDs of the current device if its dispatcher places workgroup i on XCD i mod nx (MI355X: 8), else 0; probed once per device
        int persist_xcds()
        {
            static int cached[64];
            static bool known[64] = {false};
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
            if (known[dev]) return cached[dev];
            int nx = 0;
            uint32_t *d = nullptr, h[128];
            if (hipMalloc(&d, sizeof(h)) == hipSuccess)
            {
                hipLaunchKernelGGL(persist_probe_xcc, dim3(128), dim3(64), 0, 0, d);
                if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess)
                {
                    uint32_t mx = 0;
                    for (uint32_t v : h) mx = v > mx ? v : mx;
                    nx = (int)mx + 1;
                    for (uint32_t i = 0; i < 128; i++)
                        if (h[i] != i % (uint32_t)nx) nx = 0;
                }
                (void)hipFree(d);
            }
            cached[dev] = nx;
            known[dev]  = true;
            return nx;
        }
        struct PersistForm
        {
            int nw, cpw; // wavefronts per workgroup, columns per wavefront; nw == 0: no in-launch form fits this device
            int xcds;    // > 0: the launch is xcds x G workgroups and only those on XCD 0 work
            size_t lds;
            uint32_t G;
        };
        size_t persist_lds_bytes(int ptc, uint32_t n, uint32_t maxdim) { return 8 * ((size_t)ptc * (maxdim | 1u) + 2 * (size_t)maxdim) + 8 * (size_t)(n + 1); }
        inline uint32_t persist_colld(uint32_t maxdim) { return (maxdim + 3u) & ~1u; } // a column's granules + {fresh, tail} squared norms
        template <int NW, int CPW, bool ONEXCD> bool persist_fits(uint32_t G, size_t lds, int xcds)
        {
            // The workgroups of the launch wait for each other, so ALL G of them must be resident at once — on the ONE XCD they run on, in the
            // one-XCD form: checked against what THIS device can hold (occupancy query x CU count; one workgroup fewer per CU than the query says
            // where more than one fits — MI355X_MICROARCH.md, "Residency and cooperative launch").  A plain launch has the same residency as a
            // cooperative one (same guide), and every spin is bounded, so a partitioned or busy device costs a fallback, never a hang.
            int dev = 0, cus = 0, per_cu = 0;
            if (lds > kMaxLdsBytes || G > 256u) return false;
            if (lds > 64 * 1024 &&
                hipFuncSetAttribute(reinterpret_cast<const void *>(fast_level_persist<NW, CPW, ONEXCD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return false;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fast_level_persist<NW, CPW, ONEXCD>, 64 * NW, lds) != hipSuccess)
                return false;
            if (ONEXCD) cus /= xcds;
            static const int margin = std::getenv("LEXLS_PERSIST_MARGIN") ? std::atoi(std::getenv("LEXLS_PERSIST_MARGIN")) : 1;
            const long resident     = (long)cus * (per_cu > 1 ? per_cu - margin : per_cu);
            if (std::getenv("LEXLS_LARGE_DEBUG")) std::fprintf(stderr, "lqr_large: form nw=%d cpw=%d one_xcd=%d: G=%u, %d per CU x %d CUs\n", NW, CPW, (int)ONEXCD, G, per_cu, cus);
            return per_cu >= 1 && resident >= (long)G;
        }
// the instantiated forms: (wavefronts per workgroup, columns per wavefront)
#define LEXLS_PERSIST_FORMS(X) X(4, 1) X(4, 2) X(4, 4) X(8, 1) X(8, 2) X(16, 1)
        PersistForm choose_persist_form(uint32_t n, uint32_t maxdim)
        {
            static const char *want   = std::getenv("LEXLS_PERSIST_FORM"); // "nw,cpw"
            static const int want_one = std::getenv("LEXLS_LARGE_ONE_XCD") ? std::atoi(std::getenv("LEXLS_LARGE_ONE_XCD")) : 0; // (measured: no gain, see the kernel's comment)
            const int xcds            = want_one ? persist_xcds() : 0;
            PersistForm f{0, 0, 0, 0, 0};
            if (n + 1u >= 0xFFFFFu || maxdim >= 0xFFFFu) return f; // (the record's 20-bit position and 16-bit tag fields)
            auto try_form = [&](int nw, int cpw, bool one) {
                if (f.nw) return;
                const uint32_t ptc = (uint32_t)(nw * cpw), G = (n + ptc) / ptc;
                const size_t lds   = persist_lds_bytes((int)ptc, n, maxdim);
                bool ok            = false;
#define LEXLS_X(NW_, CPW_) \
    if (nw == NW_ && cpw == CPW_) ok = one ? persist_fits<NW_, CPW_, true>(G, lds, xcds) : persist_fits<NW_, CPW_, false>(G, lds, xcds);
                LEXLS_PERSIST_FORMS(LEXLS_X)
#undef LEXLS_X
                if (ok) f = PersistForm{nw, cpw, one ? xcds : 0, lds, G};
            };
            int wn = 0, wc = 0;
            if (want && std::sscanf(want, "%d,%d", &wn, &wc) != 2) wn = wc = 0;
            if (xcds > 1)
            {
                if (wn) try_form(wn, wc, true);
                try_form(4, 4, true);
                try_form(4, 2, true);
                try_form(4, 1, true);
            }
            if (wn) try_form(wn, wc, false);
            try_form(4, 1, false);
            return f;
        }
        template <typename... Args> void launch_persist(const PersistForm &pf, size_t lds, hipStream_t s, Args... args)
        {
            const dim3 grid(pf.xcds ? (uint32_t)pf.xcds * pf.G : pf.G);
#define LEXLS_X(NW_, CPW_)                                                                                                      \
    if (pf.nw == NW_ && pf.cpw == CPW_)                                                                                         \
    {                                                                                                                           \
        if (pf.xcds)                                                                                                            \
            hipLaunchKernelGGL((fast_level_persist<NW_, CPW_, true>), grid, dim3(64 * NW_), lds, s, args..., pf.G);             \
        else                                                                                                                    \
            hipLaunchKernelGGL((fast_level_persist<NW_, CPW_, false>), grid, dim3(64 * NW_), lds, s, args..., pf.G);            \
    }
            LEXLS_PERSIST_FORMS(LEXLS_X)
#undef LEXLS_X
        }
    } // namespace

    size_t large_fast_workspace_bytes(uint32_t batch, uint32_t n, uint32_t cap, uint32_t maxdim)
    {
        const size_t ps = (size_t)cap * (n + 1);
        const size_t G = (n + PTC_MIN) / PTC_MIN; // most workgroups of the one-launch-per-level forms
        return 8 * ((size_t)batch * ps + 3 * (size_t)batch * n + (size_t)batch * maxdim * maxdim) + 4 * 2 * (size_t)batch * (n + 1) + 2 * sizeof(LargeState) * (size_t)batch + 256 +
               sizeof(PersistCtl) + persist_mailbox_bytes((uint32_t)G) + 16 * 2 * G * (size_t)((maxdim + 3u) & ~1u) + 64;
    }

    /// the fast large path (see the comment above fast_level_begin); gemm_only_mfma: the bit-exact multi-launch path with its trailing update on the matrix cores
    hipError_t launch_lqr_large_fast(const LseArgs &a, const uint32_t *h_level_max, uint32_t h_rows_max, void *d_ws, hipStream_t s)
    {
        const uint32_t B = a.batch, n = a.nVar, cap = a.cap;
        uint32_t maxdim  = 0;
        for (uint32_t k = 0; k < a.nObj; k++) maxdim = h_level_max[k] > maxdim ? h_level_max[k] : maxdim;
        const size_t ps = (size_t)cap * (n + 1);
        FastBuffers fb;
        char *w      = static_cast<char *>(d_ws);
        fb.W[0]      = a.fac;
        fb.W[1]      = reinterpret_cast<double *>(w);
        w += 8 * (size_t)B * ps;
        fb.norms[0] = reinterpret_cast<double *>(w);
        w += 8 * (size_t)B * n;
        fb.norms[1] = reinterpret_cast<double *>(w);
        w += 8 * (size_t)B * n;
        fb.D = reinterpret_cast<double *>(w);
        w += 8 * (size_t)B * n;
        fb.E = reinterpret_cast<double *>(w);
        w += 8 * (size_t)B * maxdim * maxdim;
        fb.eld   = maxdim;
        fb.st[0] = reinterpret_cast<LargeState *>(w);
        w += sizeof(LargeState) * (size_t)B;
        fb.st[1] = reinterpret_cast<LargeState *>(w);
        w += sizeof(LargeState) * (size_t)B;
        fb.pos[0] = reinterpret_cast<uint32_t *>(w);
        w += 4 * (size_t)B * (n + 1);
        fb.pos[1] = reinterpret_cast<uint32_t *>(w);
        w += 4 * (size_t)B * (n + 1);
        w = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(w) + 63) & ~(uintptr_t)63);
        PersistCtl *ctl = reinterpret_cast<PersistCtl *>(w);
        w += sizeof(PersistCtl);
        // single problems: the pivots of a level in ONE launch (fast_level_persist).  Its hand-offs spin (bounded); a launch whose spins ran out
        // raises `abort`, ends, and the level is redone with a launch per pivot (LEXLS_LARGE_PERSIST=0: always a launch per pivot; =2: raise
        // `abort` at once, for the tests).  Form of the launch: on ONE XCD when the device's workgroup placement was seen to be round robin
        // (persist_xcds) and that XCD can hold all workgroups (LEXLS_LARGE_ONE_XCD=0: never); LEXLS_PERSIST_FORM="nw,cpw" picks another instantiation.
        const PersistForm pf = choose_persist_form(n, maxdim);
        const uint32_t G     = pf.nw ? pf.G : 1u;
        if (std::getenv("LEXLS_LARGE_DEBUG")) std::fprintf(stderr, "lqr_large: in-launch form nw=%d cpw=%d xcds=%d G=%u lds=%zu\n", pf.nw, pf.cpw, pf.xcds, pf.G, pf.lds);
        PersistCand *cand    = reinterpret_cast<PersistCand *>(w);
        w += persist_mailbox_bytes((n + PTC_MIN) / PTC_MIN);
        double *colbuf           = reinterpret_cast<double *>(w);
        const size_t persist_lds = pf.lds;
        bool persist = B == 1 && pf.nw != 0 && a.skip == nullptr && !(std::getenv("LEXLS_LARGE_PERSIST") && std::atoi(std::getenv("LEXLS_LARGE_PERSIST")) == 0);
        const bool persist_test_abort = persist && std::getenv("LEXLS_LARGE_PERSIST") && std::atoi(std::getenv("LEXLS_LARGE_PERSIST")) == 2;

        hipError_t e         = hipSuccess;
        const size_t step_lds = 16 * (size_t)maxdim;
        const LargeLds lds    = large_lds_bytes(n, maxdim);
        if (step_lds > kMaxLdsBytes) return hipErrorInvalidValue;
        if (step_lds > 64 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void *>(fast_step), hipFuncAttributeMaxDynamicSharedMemorySize, (int)step_lds);
        if (e == hipSuccess && lds.trsm > 64 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void *>(large_trsm), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds.trsm);
        if (e != hipSuccess) return e;

        hipLaunchKernelGGL(large_init, dim3(64, B), dim3(256), 0, s, a, fb.st[0]);
        std::vector<LargeState> host(B);
        bool all_exhausted = false;
        uint32_t cur = 0, pp = 0;
        for (uint32_t level = 0; level < a.nObj; level++)
        {
            hipLaunchKernelGGL(fast_level_begin, dim3((n + 4) / 4, B), dim3(256), 0, s, a, fb, cur, pp, level);
            bool run_steps = false;
            if (!all_exhausted && persist && h_level_max[level] > 0) // (an empty level has no pivot: nothing to launch, no state parity to flip)
            {
                // the tags restart with every level (and every call): records and column granules of earlier pivots must not match them
                e = hipMemsetAsync(ctl, 0, sizeof(PersistCtl) + persist_mailbox_bytes((n + PTC_MIN) / PTC_MIN) + 16 * 2 * (size_t)G * persist_colld(maxdim), s);
                if (e != hipSuccess) return e;
                if (persist_test_abort)
                {
                    const uint32_t one = 1u;
                    e = hipMemcpyAsync(&ctl->abort, &one, 4, hipMemcpyHostToDevice, s);
                    if (e != hipSuccess) return e;
                }
                launch_persist(pf, persist_lds, s, a, fb, ctl, cand, colbuf, persist_colld(maxdim), cur, pp, level);
                PersistCtl hc;
                e = hipMemcpyAsync(&hc, ctl, sizeof(hc), hipMemcpyDeviceToHost, s);
                if (e != hipSuccess) return e;
                e = hipStreamSynchronize(s);
                if (e != hipSuccess) return e;
                if (hc.abort && std::getenv("LEXLS_LARGE_DEBUG")) std::fprintf(stderr, "lqr_large: level %u: the in-launch form gave up, a launch per pivot instead\n", level);
                if (hc.abort) // a hand-off ran out of spins (workgroups not co-resident?): nothing of the level was committed — a launch per pivot instead
                    run_steps = true;
                else
                    pp ^= 1u;
            }
            else if (!all_exhausted)
                run_steps = true;
            if (run_steps)
                for (uint32_t counter = 0; counter < h_level_max[level]; counter++)
                {
                    hipLaunchKernelGGL(fast_step, dim3((n + FTC) / FTC, B), dim3(FNT), step_lds, s, a, fb, cur, pp, counter);
                    pp ^= 1u;
                }
            hipLaunchKernelGGL(fast_level_end, dim3((h_rows_max + 1023) / 1024, n + 1, B), dim3(256), 0, s, a, fb, cur, pp, level);
            cur ^= 1u;
            hipLaunchKernelGGL(fast_level_commit, dim3((B + 63) / 64), dim3(64), 0, s, a, fb, pp);
            if (level + 1 < a.nObj && h_rows_max > 0 && h_level_max[level] > 0) // (a level that is empty in every problem has rank 0: no Gauss step)
            {
                LseArgs ac = a;
                ac.fac     = fb.W[cur];
                if (h_level_max[level] <= 1024)
                    hipLaunchKernelGGL(large_trsm_cols, dim3((h_rows_max + TRB - 1) / TRB, B), dim3(((h_level_max[level] + 63) / 64) * 64), 0, s, ac, fb.st[pp], level);
                else
                    hipLaunchKernelGGL(large_trsm, dim3((h_rows_max + 63) / 64, B), dim3(64), lds.trsm, s, ac, fb.st[pp], level);
                hipLaunchKernelGGL(large_gemm_mfma, dim3((h_rows_max + GBM - 1) / GBM, (n + GBN) / GBN, B), dim3(256), 0, s, ac, fb.st[pp], level);
            }
            e = hipGetLastError();
            if (e != hipSuccess) return e;
            if (!all_exhausted && level + 1 < a.nObj)
            {
                e = hipMemcpyAsync(host.data(), fb.st[pp], sizeof(LargeState) * B, hipMemcpyDeviceToHost, s);
                if (e != hipSuccess) return e;
                e = hipStreamSynchronize(s);
                if (e != hipSuccess) return e;
                all_exhausted = (a.skip == nullptr);
                for (uint32_t b = 0; b < B && all_exhausted; b++)
                    if (!host[b].exhausted) all_exhausted = false;
            }
        }
        if (cur != 0) hipLaunchKernelGGL(fast_copy_back, dim3(64, B), dim3(256), 0, s, a, fb, cur);
        hipLaunchKernelGGL(large_finish, dim3((B + 63) / 64), dim3(64), 0, s, a, fb.st[pp]);
        return hipGetLastError();
    }
} // namespace lexls
