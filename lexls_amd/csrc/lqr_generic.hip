// Generic lexicographic-QR kernels: ONE WORKGROUP PER PROBLEM, any shape.
//
// This is the shape-agnostic path of liblexls_hip: it factors and solves a problem of any size
// (matrix staged in LDS when it fits in the CU's 160 KiB, otherwise worked on in HBM/L2) and is the
// fallback behind the shape-specialised kernels.  It follows the reference algorithm statement by
// statement — LexLSE::factorize() lexlse.h:117-506, solve() :1015-1045 — and evaluates every
// reduction in the order fixed by the arithmetic contract in oracle/lexlse_oracle.h (ascending fma
// chains, first-occurrence argmax, IEEE division/sqrt), so its results are bit-identical to the
// CPU oracle's.  Parallelism inside a problem: columns across lanes for norms / Householder
// application (each lane owns whole columns, so every dot product is a lane-local ordered chain,
// no cross-lane reduction), rows across lanes for the column swap and the Gauss TRSM, elements
// across lanes for the trailing update.
//
// LDS image: column-major with an ODD leading dimension ldp so that "lane j reads (i, j)" hits 32
// distinct 8-byte bank pairs (ds_read_b64: bank = (addr/4) % 64, stride 2*ldp dwords).
#include "lexls_kernels.h"
#include "lexls_launch.h"
#include "lqr_wave_common.h"
#include "lexls_regularize.h"
#include "lexls_sweep_impl.h"

#include <cfloat>
#include <cstdlib>

namespace lexls
{
    namespace
    {
// Diagnostic build only (-DLEXLS_GENERIC_STAMPS): thread 0 of lqr_generic_kernel accumulates shader-clock totals per phase and leaves them in
// the (otherwise unused) multiplier buffer; the product build contains no stamp.  Phases: 0 stage + init + fixed variables, 1 level norms,
// 2 pivot search, 3 fresh norm + reflector scalars, 4 column swap + essential part, 5 reflector application + down-date,
// 6 regularization + Gauss step, 7 results + factor store, 8 solve, 9 the dot-product phase of 5 (up to its barrier)
#ifdef LEXLS_GENERIC_STAMPS
#define GSTAMP_DECL                 \
    unsigned long long gs_acc[10];   \
    for (int i_ = 0; i_ < 10; i_++) gs_acc[i_] = 0; \
    unsigned long long gs_t0 = clock64();
#define GSTAMP(i)                                  \
    {                                              \
        const unsigned long long t_ = clock64();   \
        gs_acc[i] += t_ - gs_t0;                   \
        gs_t0 = t_;                                \
    }
#define GSTAMP_WRITE \
    if (tid == 0)    \
        for (int i_ = 0; i_ < 10; i_++) a.lambda[(size_t)b * (n + cap) + i_] = (double)gs_acc[i_];
#else
#define GSTAMP_DECL
#define GSTAMP(i)
#define GSTAMP_WRITE
#endif

        /// block-wide "first index of the maximum" (every thread passes its best candidate) with ONE barrier: the wavefront's maximum by DPP (wave_max), the first position that attains it by a minimum over the
        /// lanes that hold it, then the NT/64 wavefront results through LDS; the slots alternate with `parity` (a pivot's slots are not
        /// rewritten before the pivot after the next one, several barriers later).  rv / ri hold at least 32 entries (NT >= 64).
        template <int NT>
        __device__ __forceinline__ uint32_t block_argmax_fast(double v, uint32_t idx, double *rv, uint32_t *ri, uint32_t tid, uint32_t parity)
        {
            const double m = wave_max(v);
            unsigned c     = row_min16(v == m ? idx : 0xffffffffu);
            {
                const unsigned c0 = (unsigned)__builtin_amdgcn_readlane((int)c, 0), c1 = (unsigned)__builtin_amdgcn_readlane((int)c, 16);
                const unsigned c2 = (unsigned)__builtin_amdgcn_readlane((int)c, 32), c3 = (unsigned)__builtin_amdgcn_readlane((int)c, 48);
                const unsigned a01 = c0 < c1 ? c0 : c1, a23 = c2 < c3 ? c2 : c3;
                c = a01 < a23 ? a01 : a23;
            }
            if (NT == 64) return c;
            const uint32_t base = parity * 16u;
            if ((tid & 63u) == 0)
            {
                rv[base + (tid >> 6)] = m;
                ri[base + (tid >> 6)] = c;
            }
            __syncthreads();
            double fm   = rv[base];
            uint32_t fi = ri[base];
#pragma unroll
            for (uint32_t w = 1; w < (uint32_t)(NT / 64); w++)
            {
                const double v2   = rv[base + w];
                const uint32_t i2 = ri[base + w];
                if (v2 > fm || (v2 == fm && i2 < fi))
                {
                    fm = v2;
                    fi = i2;
                }
            }
            return fi;
        }

        /// s = fma(p_i, p_i, s) for i ascending.  The reads of a chunk of U are issued together and the next chunk is fetched (into the
        /// other of two register sets) before the current one is consumed, so the chain runs at the latency of the fused multiply-add
        /// instead of a memory round trip per term (same order, same result)
        template <int U>
        __device__ __forceinline__ double chain_square(const double *p, uint32_t len, double s)
        {
            uint32_t i = 0;
            if (len >= (uint32_t)U)
            {
                double v0[U], v1[U];
#pragma unroll
                for (int u = 0; u < U; u++) v0[u] = p[u];
                i = U;
                while (true)
                {
                    const bool more1 = i + U <= len;
                    if (more1)
                    {
#pragma unroll
                        for (int u = 0; u < U; u++) v1[u] = p[i + u];
                        i += U;
                    }
#pragma unroll
                    for (int u = 0; u < U; u++) s = dfma(v0[u], v0[u], s);
                    if (!more1) break;
                    const bool more0 = i + U <= len;
                    if (more0)
                    {
#pragma unroll
                        for (int u = 0; u < U; u++) v0[u] = p[i + u];
                        i += U;
                    }
#pragma unroll
                    for (int u = 0; u < U; u++) s = dfma(v1[u], v1[u], s);
                    if (!more0) break;
                }
            }
            for (; i < len; i++) s = dfma(p[i], p[i], s);
            return s;
        }
        /// the two chains of a pivot column in one pass: fresh = sum_{i<R} c_i^2 and tail = sum_{1<=i<R} c_i^2, each ascending from 0
        template <int U>
        __device__ __forceinline__ void chain_square_pair(const double *c, uint32_t R, double &fresh, double &tail)
        {
            double f = dfma(c[0], c[0], 0.0), t = 0.0;
            uint32_t i = 1;
            if (R >= 1u + (uint32_t)U)
            {
                double v0[U], v1[U];
#pragma unroll
                for (int u = 0; u < U; u++) v0[u] = c[1 + u];
                i = 1 + U;
                while (true)
                {
                    const bool more1 = i + U <= R;
                    if (more1)
                    {
#pragma unroll
                        for (int u = 0; u < U; u++) v1[u] = c[i + u];
                        i += U;
                    }
#pragma unroll
                    for (int u = 0; u < U; u++)
                    {
                        f = dfma(v0[u], v0[u], f);
                        t = dfma(v0[u], v0[u], t);
                    }
                    if (!more1) break;
                    const bool more0 = i + U <= R;
                    if (more0)
                    {
#pragma unroll
                        for (int u = 0; u < U; u++) v0[u] = c[i + u];
                        i += U;
                    }
#pragma unroll
                    for (int u = 0; u < U; u++)
                    {
                        f = dfma(v1[u], v1[u], f);
                        t = dfma(v1[u], v1[u], t);
                    }
                    if (!more0) break;
                }
            }
            for (; i < R; i++)
            {
                f = dfma(c[i], c[i], f);
                t = dfma(c[i], c[i], t);
            }
            fresh = f;
            tail  = t;
        }
        /// s = fma(a_i, b_i, s) for i ascending, reads batched and prefetched as in chain_square
        template <int U>
        __device__ __forceinline__ double chain_dot(const double *a, const double *b, uint32_t len, double s)
        {
            uint32_t i = 0;
            if (len >= (uint32_t)U)
            {
                double x0[U], y0[U], x1[U], y1[U];
#pragma unroll
                for (int u = 0; u < U; u++)
                {
                    x0[u] = a[u];
                    y0[u] = b[u];
                }
                i = U;
                while (true)
                {
                    const bool more1 = i + U <= len;
                    if (more1)
                    {
#pragma unroll
                        for (int u = 0; u < U; u++)
                        {
                            x1[u] = a[i + u];
                            y1[u] = b[i + u];
                        }
                        i += U;
                    }
#pragma unroll
                    for (int u = 0; u < U; u++) s = dfma(x0[u], y0[u], s);
                    if (!more1) break;
                    const bool more0 = i + U <= len;
                    if (more0)
                    {
#pragma unroll
                        for (int u = 0; u < U; u++)
                        {
                            x0[u] = a[i + u];
                            y0[u] = b[i + u];
                        }
                        i += U;
                    }
#pragma unroll
                    for (int u = 0; u < U; u++) s = dfma(x1[u], y1[u], s);
                    if (!more0) break;
                }
            }
            for (; i < len; i++) s = dfma(a[i], b[i], s);
            return s;
        }

        struct Shared
        {
            double tau, beta, den;
            uint32_t piv;
            int stop, degenerate;
        };

        /// v <- Q_k v (Householder sequence H_0 ... H_{r-1}, last reflector first; lexlse.h:1576-1578).
        /// v points at the level's first entry (LDS); W is the factor.  Block-wide.
        template <int NT>
        __device__ void apply_q_block(const double *W, size_t ld, const double *hh, uint32_t F, uint32_t Fc, uint32_t dim, uint32_t rank, double *v,
                                      double *bcast, uint32_t tid)
        {
            for (uint32_t j = rank; j--;)
            {
                const uint32_t rows = dim - j;
                const double tau    = hh[F + j];
                const double *ess   = W + (F + j + 1) + (size_t)(Fc + j) * ld;
                if (rows == 1)
                {
                    if (tid == 0) v[j] *= (1.0 - tau);
                    __syncthreads();
                }
                else if (tau != 0.0)
                {
                    if (tid == 0)
                    {
                        double t = chain_dot<8>(ess, v + j + 1, rows - 1, 0.0);
                        t += v[j];
                        *bcast = t;
                    }
                    __syncthreads();
                    const double t = *bcast;
                    for (uint32_t i = tid; i < rows; i += NT)
                    {
                        if (i == 0)
                            v[j] = dfma(-tau, t, v[j]);
                        else
                            v[j + i] = dfma(-(tau * ess[i - 1]), t, v[j + i]);
                    }
                    __syncthreads();
                }
            }
        }

        /// apply_q_block for ONE wavefront and a level of at most 64 rows: lane i keeps v[i] in a register, the ordered dot product of
        /// a reflector walks the lanes with v_readlane (a few cycles per term instead of a dependent LDS round trip), the update is
        /// lane-local and there is no barrier inside the sequence.  Same chains, same order, same results.
        __device__ __forceinline__ void apply_q_wave(const double *W, size_t ld, const double *hh, uint32_t F, uint32_t Fc, uint32_t dim, uint32_t rank, double *v,
                                                     uint32_t lane)
        {
            double vi = lane < dim ? v[lane] : 0.0;
            for (uint32_t j = rank; j--;)
            {
                const uint32_t rows = dim - j;
                const double tau    = hh[F + j];
                if (rows == 1)
                {
                    if (lane == j) vi *= (1.0 - tau);
                }
                else if (tau != 0.0)
                {
                    const bool tail = lane > j && lane < dim;
                    const double e  = tail ? W[(F + lane) + (size_t)(Fc + j) * ld] : 0.0; // essential part: entry of row j + i sits in lane j + i
                    double t        = 0.0;
                    for (uint32_t i = 1; i < rows; i++)
                    {
                        const int l     = (int)(j + i);
                        const double ei = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(e), l), __builtin_amdgcn_readlane(__double2loint(e), l));
                        const double xi = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(vi), l), __builtin_amdgcn_readlane(__double2loint(vi), l));
                        t               = dfma(ei, xi, t);
                    }
                    t += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(vi), (int)j), __builtin_amdgcn_readlane(__double2loint(vi), (int)j));
                    if (lane == j)
                        vi = dfma(-tau, t, vi);
                    else if (tail)
                        vi = dfma(-(tau * e), t, vi);
                }
            }
            if (lane < dim) v[lane] = vi;
            __syncthreads();
        }

        // -----------------------------------------------------------------------------------------
        // factorize (+ solve)
        // -----------------------------------------------------------------------------------------
        /// x = P x (lexlse.h:1044): the transpositions k <-> perm[k], last first, on x by position — read the other way round: variable i follows
        /// them first to last from position i to its final position p, and x_i is what sits there.  Every thread for its own variables, on indices
        /// only (one thread swapping entries behind a load of perm[k] each was 128 of the 150 us of configs[1]'s solve).  perm_l: LDS scratch of T
        /// words, 16-BYTE ALIGNED (it is read four words at a time), that nothing else uses while this runs; xs: x by position (LDS)
        template <int NT>
        __device__ __forceinline__ void apply_permutation_by_following(const LseArgs &a, uint32_t b, const uint32_t *perm, uint32_t T, const double *xs, uint32_t *perm_l, uint32_t tid)
        {
            const uint32_t n = a.nVar;
            for (uint32_t k = tid; k < T; k += NT) perm_l[k] = perm[k];
            __syncthreads();
            // (UV variables per thread and pass: independent index chains side by side)
            auto follow = [&](auto uv_c, uint32_t i0) __attribute__((always_inline)) {
                constexpr int UV = decltype(uv_c)::value;
                uint32_t p[UV];
#pragma unroll
                for (int u = 0; u < UV; u++) p[u] = i0 + u * NT;
                for (uint32_t k = 0; k + 4 <= T; k += 4)
                {
                    const uint4 pk4      = *reinterpret_cast<const uint4 *>(perm_l + k);
                    const uint32_t pk[4] = {pk4.x, pk4.y, pk4.z, pk4.w};
#pragma unroll
                    for (int kk = 0; kk < 4; kk++)
#pragma unroll
                        for (int u = 0; u < UV; u++) p[u] = (p[u] == k + kk) ? pk[kk] : ((p[u] == pk[kk]) ? k + kk : p[u]);
                }
                for (uint32_t k = T & ~3u; k < T; k++)
                {
                    const uint32_t pk = perm_l[k];
#pragma unroll
                    for (int u = 0; u < UV; u++) p[u] = (p[u] == k) ? pk : ((p[u] == pk) ? k : p[u]);
                }
#pragma unroll
                for (int u = 0; u < UV; u++)
                    if (i0 + u * NT < n) a.x[(size_t)b * n + i0 + u * NT] = xs[p[u]];
            };
            const uint32_t rows = (n + NT - 1) / NT; // variables per thread (the last pass may reach beyond n: those are not stored)
            uint32_t done       = 0;
            for (; rows - done >= 4; done += 4) follow(std::integral_constant<int, 4>{}, tid + done * NT);
            switch (rows - done)
            {
            case 3: follow(std::integral_constant<int, 3>{}, tid + done * NT); break;
            case 2: follow(std::integral_constant<int, 2>{}, tid + done * NT); break;
            case 1: follow(std::integral_constant<int, 1>{}, tid + done * NT); break;
            default: break;
            }
        }

        template <int NT, bool LDSMAT>
        __global__ __launch_bounds__(NT) void lqr_generic_kernel(LseArgs a, int write_factor, int do_solve)
        {
            extern __shared__ double smem[];
            const uint32_t b = blockIdx.x, tid = threadIdx.x;
            const uint32_t n = a.nVar, cap = a.cap, nObj = a.nObj;
            const size_t pstride = (size_t)cap * (n + 1);
            if (a.skip && a.skip[b]) return; // uniform per workgroup
            // terms per read batch of the ordered chains: the one-wavefront form serves large batches of small problems, where registers
            // are occupancy (16 -> 157 VGPRs, 3 wavefronts per SIMD: measured slower there); the wide forms run one workgroup per CU
            constexpr int CH = NT >= 256 ? 16 : 4;
            GSTAMP_DECL

            // ---- LDS carve-up (doubles first) ----
            double *W;
            size_t ld;
            double *p = smem;
            if (LDSMAT)
            {
                W  = p;
                ld = a.ldp;
                p += (size_t)a.ldp * (n + 1);
            }
            else
            {
                W  = a.fac + b * pstride;
                ld = cap;
            }
            double *norms  = p;
            double *xs     = norms + n;
            double *red_v  = xs + n;
            Shared *sh     = reinterpret_cast<Shared *>(red_v + NT);
            uint32_t *red_i  = reinterpret_cast<uint32_t *>(sh + 1);
            uint32_t *perm_s = red_i + NT;
            uint32_t *rk_s   = perm_s + n;
            uint32_t *fc_s   = rk_s + nObj;
            uint32_t *fr_s   = fc_s + nObj;

            const uint32_t *dims = a.dims + (size_t)b * nObj;
            uint32_t M           = 0;
            for (uint32_t k = 0; k < nObj; k++)
            {
                if (tid == 0)
                {
                    fr_s[k] = M;
                    rk_s[k] = 0;
                    fc_s[k] = 0;
                }
                M += dims[k];
            }

            // ---- fixed variables (lexlse.h:132-143): the column transpositions, and — when the matrix is staged into LDS — the column
            // order they leave behind, so that the staging pass below places the columns directly (one pass instead of one per variable)
            const uint32_t nf = a.nfixed ? a.nfixed[b] : 0;
            for (uint32_t i = tid; i < n; i += NT) perm_s[i] = i;
            const uint32_t *src = nullptr;
            if (nf > 0)
            {
                uint32_t *order = reinterpret_cast<uint32_t *>(norms); // norms is free until the level loop: n + n entries
                uint32_t *fidx  = order + n;                             // scratch copy of fixed_var_index
                uint32_t *where = reinterpret_cast<uint32_t *>(xs);    // xs is initialised after the staging pass: slot that holds a value
                for (uint32_t k = tid; k < nf; k += NT) fidx[k] = a.fixed_idx[(size_t)b * n + k];
                for (uint32_t v = tid; v < n; v += NT) where[v] = 0xffffffffu;
                __syncthreads();
                if (tid == 0)
                {
                    // lexlse.h:146-153: slot k takes its variable; the later slot that named variable k is redirected to the column that
                    // variable k was swapped into.  With distinct indices (the normal case) "the later slot" is a table look-up
                    bool distinct = true;
                    for (uint32_t i = 0; i < nf; i++)
                    {
                        const uint32_t v = fidx[i];
                        if (where[v] != 0xffffffffu) distinct = false;
                        else where[v] = i;
                    }
                    for (uint32_t k = 0; k < nf; k++)
                    {
                        const uint32_t coeff = fidx[k];
                        perm_s[k]            = coeff;
                        if (distinct)
                        {
                            const uint32_t i = where[k];
                            if (i != 0xffffffffu && i > k)
                            {
                                fidx[i]      = coeff;
                                where[coeff] = i;
                            }
                            continue;
                        }
                        for (uint32_t i = k + 1; i < nf; i++)
                        {
                            if (fidx[i] == k)
                            {
                                fidx[i] = coeff;
                                break;
                            }
                        }
                    }
                    if (LDSMAT)
                    {
                        for (uint32_t j = 0; j < n; j++) order[j] = j;
                        for (uint32_t k = 0; k < nf; k++) // swapping columns k and perm[k] = swapping where they come from
                        {
                            const uint32_t c = perm_s[k], t = order[k];
                            order[k]         = order[c];
                            order[c]         = t;
                        }
                    }
                }
                if (LDSMAT) src = reinterpret_cast<const uint32_t *>(norms);
                __syncthreads();
            }

            // ---- stage the problem (coalesced down the columns) ----
            const double *in = a.in + b * pstride;
            if (LDSMAT || in != W)
            {
                // every thread busy and eight independent loads in flight per thread (a column at a time leaves NT - M threads idle and
                // one global round trip per column on the critical path)
                const uint32_t total = M * (n + 1);
                constexpr int SU = 8;
                for (uint32_t e0 = tid; e0 < total; e0 += SU * NT)
                {
                    double v[SU];
                    uint32_t dst[SU];
#pragma unroll
                    for (int u = 0; u < SU; u++)
                    {
                        const uint32_t e = e0 + (uint32_t)u * NT;
                        const uint32_t j = e < total ? e / M : 0, i = e < total ? e - j * M : 0;
                        const size_t sj  = (src && j < n) ? src[j] : j;
                        v[u]             = e < total ? in[i + sj * cap] : 0.0;
                        dst[u]           = e < total ? (uint32_t)(i + j * ld) : 0xffffffffu;
                    }
#pragma unroll
                    for (int u = 0; u < SU; u++)
                        if (dst[u] != 0xffffffffu) W[dst[u]] = v[u];
                }
            }
            double *hh = a.hh + (size_t)b * cap;
            for (uint32_t i = tid; i < cap; i += NT) hh[i] = 0.0; // initialize(), lexlse.h:1683
            for (uint32_t i = tid; i < n; i += NT) xs[i] = (i < nf) ? a.fixed_val[(size_t)b * n + i] : 0.0;
            if (a.reg_type) // initialize(): null_space.setZero() (lexlse.h:1686)
            {
                double *NS = a.reg_scratch + (size_t)b * reg_scratch_doubles(n);
                for (uint32_t e = tid; e < n * (n + 1); e += NT) NS[e] = 0.0;
            }
            double *mu_x = nullptr, *mu_res = nullptr; // by-products of the experimental type 7 (lexlse.h:96-99, zeroed by initialize() :1687-1689)
            if (a.reg_type == 7)
            {
                mu_x   = a.reg_mu + (size_t)b * reg_mu_doubles(n, nObj, cap);
                mu_res = mu_x + 2 * (size_t)nObj * n;
                for (size_t e = tid; e < reg_mu_doubles(n, nObj, cap); e += NT) mu_x[e] = 0.0;
            }
            __syncthreads();

            // ---- fixed variables: their columns go first (done by the staging pass when the matrix lives in LDS), their contribution
            // goes to the RHS (lexlse.h:132-156) ----
            if (nf > 0)
            {
                if (!LDSMAT)
                    for (uint32_t k = 0; k < nf; k++)
                    {
                        const uint32_t coeff = perm_s[k];
                        if (coeff != k)
                            for (uint32_t i = tid; i < M; i += NT)
                            {
                                const double t    = W[i + k * ld];
                                W[i + k * ld]     = W[i + coeff * ld];
                                W[i + coeff * ld] = t;
                            }
                        __syncthreads();
                    }
                for (uint32_t i = tid; i < M; i += NT)
                {
                    double s = 0.0;
                    for (uint32_t k = 0; k < nf; k++) s = dfma(W[i + k * ld], xs[k], s);
                    W[i + n * ld] -= s;
                }
                __syncthreads();
            }

            uint32_t ColIndex  = nf;
            uint32_t TotalRank = nf;
            GSTAMP(0)

            if (ColIndex < n)
            {
                for (uint32_t ObjIndex = 0; ObjIndex < nObj; ObjIndex++)
                {
                    const uint32_t F = fr_s[ObjIndex], Fc = ColIndex, dim = dims[ObjIndex];
                    if (tid == 0) fc_s[ObjIndex] = Fc;

                    if (mu_res) // lexlse.h:191: the level's right-hand side after the eliminations, before its reflectors
                        for (uint32_t i = tid; i < dim; i += NT) mu_res[F + i] = W[F + i + n * ld];

                    // initial squared column norms of the level (lexlse.h:193-196): one ordered chain per column
                    for (uint32_t k = ColIndex + tid; k < n; k += NT)
                    {
                        norms[k] = chain_square<CH>(W + F + k * ld, dim, 0.0);
                    }
                    __syncthreads();
                    GSTAMP(1)

                    for (uint32_t counter = 0; counter < dim; counter++)
                    {
                        const uint32_t row = F + counter, R = dim - counter;

                        // pivot = first maximum of the down-dated norms (lexlse.h:205-206)
                        double bv   = -INFINITY;
                        uint32_t bi = 0xffffffffu;
                        for (uint32_t k = ColIndex + tid; k < n; k += NT)
                            if (norms[k] > bv)
                            {
                                bv = norms[k];
                                bi = k;
                            }
                        const uint32_t piv = block_argmax_fast<NT>(bv, bi, red_v, red_i, tid, counter & 1u);
                        GSTAMP(2)

                        // fresh norm, rank test, Householder scalars (lexlse.h:210-217, :241)
                        if (tid == 0)
                        {
                            const double *col = W + row + piv * ld;
                            double fresh, tailSq;
                            chain_square_pair<CH>(col, R, fresh, tailSq);
                            norms[piv]     = fresh;
                            sh->stop       = fresh < a.tol;
                            sh->degenerate = 0;
                            sh->tau        = 0.0;
                            if (!sh->stop)
                            {
                                perm_s[ColIndex]       = piv;
                                const double t         = norms[ColIndex];
                                norms[ColIndex]        = norms[piv];
                                norms[piv]             = t;
                                if (R > 1)
                                {
                                    const double c0 = col[0];
                                    if (tailSq <= DBL_MIN)
                                    {
                                        sh->degenerate = 1;
                                        sh->beta       = c0;
                                    }
                                    else
                                    {
                                        double beta = sqrt(dfma(c0, c0, tailSq));
                                        if (c0 >= 0.0) beta = -beta;
                                        sh->beta = beta;
                                        sh->den  = c0 - beta;
                                        sh->tau  = (beta - c0) / beta;
                                    }
                                }
                            }
                        }
                        __syncthreads();
                        GSTAMP(3)
                        if (sh->stop) break; // uniform

                        // column swap over ALL nCtr rows (lexlse.h:222-232) fused with writing beta / the essential part
                        {
                            const double den = sh->den, beta = sh->beta;
                            const int degenerate = sh->degenerate;
                            for (uint32_t i = tid; i < M; i += NT)
                            {
                                const double a1 = W[i + ColIndex * ld];
                                const double a2 = W[i + piv * ld];
                                double newc     = a2;
                                if (R > 1)
                                {
                                    if (i == row)
                                        newc = beta;
                                    else if (i > row && i < row + R)
                                        newc = degenerate ? 0.0 : a2 / den;
                                }
                                W[i + ColIndex * ld] = newc;
                                if (piv != ColIndex) W[i + piv * ld] = a1;
                            }
                            if (a.reg_type && piv != ColIndex) // lexlse.h:229-231: the rows of the basis above this level's first column
                            {
                                double *NS = a.reg_scratch + (size_t)b * reg_scratch_doubles(n);
                                for (uint32_t i = tid; i < Fc; i += NT)
                                {
                                    const double t               = NS[i + (size_t)ColIndex * n];
                                    NS[i + (size_t)ColIndex * n] = NS[i + (size_t)piv * n];
                                    NS[i + (size_t)piv * n]      = t;
                                }
                            }
                        }
                        __syncthreads();
                        GSTAMP(4)

                        // apply H to the trailing columns incl. the RHS (lexlse.h:243-246), then down-date (:262-266)
                        {
                            const double tau  = sh->tau;
                            const double *ess = W + row + 1 + ColIndex * ld;
                            // Two phases per batch of NT columns.  A: one thread per column walks the ordered dot product (the only serial
                            // part), updates the column's pivot-row entry and down-dates its norm.  B: ALL threads apply the rank-one
                            // update to the rows below (lanes along the rows, 32 at a time; the column-per-thread form left 3/4 of the
                            // workgroup idle through half of the step).  Same operations on every entry, same results.
                            const bool reflect = R > 1 && tau != 0.0; // uniform
                            for (uint32_t base = ColIndex + 1; base <= n; base += NT)
                            {
                                const uint32_t j = base + tid;
                                if (j <= n)
                                {
                                    double *col = W + row + j * ld;
                                    if (reflect)
                                    {
                                        double tmp = chain_dot<CH>(ess, col + 1, R - 1, 0.0);
                                        tmp += col[0];
                                        col[0]     = dfma(-tau, tmp, col[0]);
                                        red_v[tid] = tmp;
                                    }
                                    if (j < n) norms[j] = dfma(-col[0], col[0], norms[j]);
                                }
                                if (reflect)
                                {
                                    __syncthreads();
                                    GSTAMP(9)
                                    const uint32_t nb = (n + 1 - base < (uint32_t)NT) ? n + 1 - base : (uint32_t)NT;
                                    const uint32_t tx = tid & 31u, ty = tid >> 5;
                                    constexpr uint32_t CG = NT / 32; // column groups
                                    for (uint32_t i = 1 + tx; i < R; i += 32)
                                    {
                                        const double te = -(tau * ess[i - 1]);
                                        uint32_t c      = ty;
                                        for (; c + 3 * CG < nb; c += 4 * CG) // four columns per trip: their reads are issued together
                                        {
                                            double *p0 = W + row + i + (base + c) * ld, *p1 = p0 + CG * ld, *p2 = p1 + CG * ld, *p3 = p2 + CG * ld;
                                            const double c0 = *p0, c1 = *p1, c2 = *p2, c3 = *p3;
                                            const double t0 = red_v[c], t1 = red_v[c + CG], t2 = red_v[c + 2 * CG], t3 = red_v[c + 3 * CG];
                                            *p0 = dfma(te, t0, c0);
                                            *p1 = dfma(te, t1, c1);
                                            *p2 = dfma(te, t2, c2);
                                            *p3 = dfma(te, t3, c3);
                                        }
                                        for (; c < nb; c += CG)
                                        {
                                            double *p0 = W + row + i + (base + c) * ld;
                                            *p0        = dfma(te, red_v[c], *p0);
                                        }
                                    }
                                    if (base + NT <= n) __syncthreads(); // red_v is rewritten by the next batch
                                }
                            }
                            if (tid == 0 && R > 1) hh[row] = tau;
                        }
                        ColIndex++;
                        __syncthreads();
                        GSTAMP(5)
                        if (ColIndex == n) break;
                    }

                    const uint32_t rank = ColIndex - Fc;
                    if (tid == 0) rk_s[ObjIndex] = rank;
                    TotalRank += rank;

                    if (a.reg_type) // lexlse.h:277-411: the level's transformed right-hand side is damped before the Gauss step
                    {
                        __syncthreads();
                        const RegLevels lv = {fr_s, fc_s, rk_s, perm_s, hh, dim};
                        regularize_level<NT>(a, b, W, ld, nf, ObjIndex, F, Fc, rank, n - ColIndex, tid, &lv);
                    }

                    // Gauss step (lexlse.h:431-471)
                    if (ObjIndex + 1 < nObj && rank > 0)
                    {
                        const uint32_t Fn = F + dim;
                        for (uint32_t i = Fn + tid; i < M; i += NT) // L <- L R^-1, one row per lane (reciprocal of the diagonal, see oracle)
                        {
                            for (uint32_t p2 = 0; p2 < rank; p2++)
                            {
                                double s = W[i + (Fc + p2) * ld];
                                for (uint32_t q = 0; q < p2; q++) s = dfma(-W[i + (Fc + q) * ld], W[F + q + (Fc + p2) * ld], s);
                                W[i + (Fc + p2) * ld] = s * (1.0 / W[F + p2 + (Fc + p2) * ld]);
                            }
                        }
                        __syncthreads();
                        const uint32_t nrows = M - Fn, ncols = n - ColIndex + 1;
                        if (nrows > 0)
                            for (uint32_t e = tid; e < nrows * ncols; e += NT) // Trailing -= L * Up, one element per lane
                            {
                                const uint32_t i = Fn + e % nrows, j = ColIndex + e / nrows;
                                double t         = W[i + j * ld];
                                for (uint32_t p2 = 0; p2 < rank; p2++) t = dfma(-W[i + (Fc + p2) * ld], W[F + p2 + j * ld], t);
                                W[i + j * ld] = t;
                            }
                        __syncthreads();
                    }

                    GSTAMP(6)
                    if (ColIndex == n) // lexlse.h:475-490
                    {
                        if (tid == 0)
                            for (uint32_t k = ObjIndex + 1; k < nObj; k++) fc_s[k] = fc_s[k - 1] + rk_s[k - 1];
                        if (mu_x) // lexlse.h:483-486
                        {
                            __syncthreads();
                            for (uint32_t k = ObjIndex + 1; k < nObj; k++)
                            {
                                for (uint32_t i = tid; i < n; i += NT)
                                {
                                    mu_x[(size_t)k * n + i]          = mu_x[(size_t)(k - 1) * n + i];
                                    mu_x[(size_t)(nObj + k) * n + i] = mu_x[(size_t)(nObj + k - 1) * n + i];
                                }
                                for (uint32_t i = tid; i < dims[k]; i += NT) mu_res[fr_s[k] + i] = -W[fr_s[k] + i + n * ld];
                                __syncthreads();
                            }
                        }
                        break;
                    }
                }
            }
            __syncthreads();

            // ---- results of factorize() ----
            for (uint32_t k = tid; k < nObj; k += NT)
            {
                a.rank[(size_t)b * nObj + k] = rk_s[k];
                a.fcol[(size_t)b * nObj + k] = fc_s[k];
            }
            for (uint32_t i = tid; i < n; i += NT) a.perm[(size_t)b * n + i] = (i < TotalRank) ? perm_s[i] : i;
            if (tid == 0) a.totalrank[b] = TotalRank;
            if (LDSMAT && write_factor)
            {
                double *out = a.fac + b * pstride;
                for (uint32_t j = 0; j <= n; j++)
                    for (uint32_t i = tid; i < M; i += NT) out[i + (size_t)j * cap] = W[i + j * ld];
            }

            GSTAMP(7)
            // ---- solve(): block back-substitution (lexlse.h:1015-1045) ----
            if (do_solve)
            {
                uint32_t acc = 0;
                for (uint32_t k = nObj; k--;)
                {
                    const uint32_t rank = rk_s[k];
                    if (rank == 0) continue;
                    const uint32_t F = fr_s[k], Fc = fc_s[k];
                    const uint32_t c0 = (acc > 0) ? fc_s[k + 1] : 0;
                    for (uint32_t i = tid; i < rank; i += NT)
                    {
                        double s = W[F + i + n * ld];
                        for (uint32_t j = 0; j < acc; j++) s = dfma(-W[F + i + (c0 + j) * ld], xs[c0 + j], s);
                        xs[Fc + i] = s;
                    }
                    __syncthreads();
                    for (uint32_t j = rank; j--;)
                    {
                        if (tid == 0) xs[Fc + j] = xs[Fc + j] / W[F + j + (Fc + j) * ld];
                        __syncthreads();
                        const double xj = xs[Fc + j];
                        for (uint32_t i = tid; i < j; i += NT) xs[Fc + i] = dfma(-W[F + i + (Fc + j) * ld], xj, xs[Fc + i]);
                        __syncthreads();
                    }
                    acc += rank;
                }
                // x = P x (lexlse.h:1044): variable i follows the transpositions, first to last, from position i to its final position
                // (solve_generic_kernel explains); the entries of x by position are only read
                __syncthreads();
                for (uint32_t i = tid; i < n; i += NT)
                {
                    uint32_t p = i;
#pragma unroll 8
                    for (uint32_t k = 0; k < TotalRank; k++)
                    {
                        const uint32_t pk = perm_s[k];
                        p                 = (p == k) ? pk : ((p == pk) ? k : p);
                    }
                    a.x[(size_t)b * n + i] = xs[p];
                }
            }
            GSTAMP(8)
            GSTAMP_WRITE
        }

        // -----------------------------------------------------------------------------------------
        // solve() alone, from the factor in HBM
        // -----------------------------------------------------------------------------------------
        /// RECIP (the large path's step-per-pivot form, whose contract is values within 1e-10): a diagonal block's 64 reciprocals are formed
        /// at once, lane-parallel, and the chain multiplies — the division sequence (~400 cycles) sat on every one of the nVar dependent steps
        template <int NT, bool RECIP = false>
        __global__ __launch_bounds__(NT) void solve_generic_kernel(LseArgs a)
        {
            extern __shared__ double smem[];
            const uint32_t b = blockIdx.x, tid = threadIdx.x;
#ifdef LEXLS_SOLVE_STAMPS
            long long sst[6] = {0, 0, 0, 0, 0, 0}, st0 = clock64();
#define SSTAMP(i) { const long long t_ = clock64(); sst[i] += t_ - st0; st0 = t_; }
#else
#define SSTAMP(i)
#endif
            const uint32_t n = a.nVar, cap = a.cap, nObj = a.nObj;
            const double *W  = a.fac + (size_t)b * cap * (n + 1);
            const size_t ld  = cap;
            double *xs       = smem;
            const uint32_t *dims = a.dims + (size_t)b * nObj;
            const uint32_t *rk = a.rank + (size_t)b * nObj, *fc = a.fcol + (size_t)b * nObj;
            const uint32_t *perm = a.perm + (size_t)b * n;
            const uint32_t nf    = a.nfixed ? a.nfixed[b] : 0;
            for (uint32_t i = tid; i < n; i += NT) xs[i] = (i < nf) ? a.fixed_val[(size_t)b * n + i] : 0.0;
            __syncthreads();
            uint32_t M = 0;
            for (uint32_t k = 0; k < nObj; k++) M += dims[k];
            uint32_t acc = 0, Fend = M;
            for (uint32_t k = nObj; k--;)
            {
                const uint32_t F = Fend - dims[k];
                Fend             = F;
                const uint32_t rank = rk[k];
                if (rank == 0) continue;
                const uint32_t Fc = fc[k];
                const uint32_t c0 = (acc > 0) ? fc[k + 1] : 0;
                for (uint32_t i = tid; i < rank; i += NT)
                {
                    double s = W[F + i + n * ld];
                    uint32_t j = 0;
                    if constexpr (NT >= 256) // (large levels: sixteen entries of the row requested together; the chain itself stays in order)
                    {
                        // (a batch = one trip to the stored factor, ~1.5 us: 64 entries per trip instead of 16 took the solve of configs[1] from
                        // 150 to NN us)
                        for (; j + 64 <= acc; j += 64)
                        {
                            double w64[64];
#pragma unroll
                            for (uint32_t u = 0; u < 64; u++) w64[u] = W[F + i + (c0 + j + u) * ld];
#pragma unroll
                            for (uint32_t u = 0; u < 64; u++) s = dfma(-w64[u], xs[c0 + j + u], s);
                        }
                        for (; j + 16 <= acc; j += 16)
                        {
                            double w16[16];
#pragma unroll
                            for (uint32_t u = 0; u < 16; u++) w16[u] = W[F + i + (c0 + j + u) * ld];
#pragma unroll
                            for (uint32_t u = 0; u < 16; u++) s = dfma(-w16[u], xs[c0 + j + u], s);
                        }
                    }
                    for (; j < acc; j++) s = dfma(-W[F + i + (c0 + j) * ld], xs[c0 + j], s);
                    xs[Fc + i] = s;
                }
                SSTAMP(0)
                if constexpr (NT >= 256)
                {
                    // Large levels: the triangular solve in blocks of 64 rows from the bottom.  ONE wavefront solves a diagonal block staged in
                    // LDS (lane = row; x_j by v_readlane: no workgroup barrier per variable), then every thread applies the block's 64 columns
                    // to a row above it.  Every x_i absorbs its columns in the same descending order as the unblocked loop: bit-identical.
                    constexpr uint32_t BS = 64, BL = 65;
                    double *Rb = xs + 2 * (size_t)n + 2; // BS x BL: the diagonal block, column-major, odd leading dimension
                    for (uint32_t jb = ((rank - 1) / BS) * BS;; jb -= BS)
                    {
                        const uint32_t nb = rank - jb < BS ? rank - jb : BS;
                        {
                            // thread = (row, column mod NT / 64) of the block: sixteen loads in flight per thread, no division per element
                            // (one element at a time behind "e % nb, e / nb" the staging took 6 us per block)
                            const uint32_t i = tid & 63u, j0 = tid >> 6;
                            constexpr uint32_t JS = NT / 64;
                            double st[BS / JS];
#pragma unroll
                            for (uint32_t u = 0; u < BS / JS; u++)
                            {
                                const uint32_t j = j0 + u * JS;
                                st[u]            = (i < nb && j < nb) ? W[F + jb + i + (Fc + jb + j) * ld] : 0.0;
                            }
#pragma unroll
                            for (uint32_t u = 0; u < BS / JS; u++)
                            {
                                const uint32_t j = j0 + u * JS;
                                if (i < nb && j < nb) Rb[i + j * BL] = st[u];
                            }
                        }
                        __syncthreads();
                        SSTAMP(1)
                        if (tid < 64)
                        {
                            // lane = row of the block, its entries ABOVE the diagonal in registers (zero from the diagonal down: the update
                            // below then needs no per-lane predicate — 64 of them, kept as 128 SGPRs, spilled), the chain fully unrolled: one
                            // step = x_j from lane j (v_readlane, constant lane) times the reciprocal diagonal (or divided), one fma for the
                            // rows above, lane j keeps x_j.  A lane's own entry is read before its step's fma can touch it.
                            double sv = tid < nb ? xs[Fc + jb + tid] : 0.0;
                            double rrow[BS];
#pragma unroll
                            for (uint32_t j = 0; j < BS; j++) rrow[j] = (tid < j && j < nb) ? Rb[tid + j * BL] : 0.0;
                            double dg = tid < nb ? Rb[tid * (BL + 1)] : 1.0; // this lane's diagonal entry (RECIP: its reciprocal)
                            if (RECIP) dg = 1.0 / dg;
                            double xfin = 0.0;
                            for_each_index<0, (int)BS>([&](auto jc) __attribute__((always_inline)) {
                                constexpr int j = (int)BS - 1 - decltype(jc)::value;
                                if ((uint32_t)j < nb) // (wave-uniform)
                                {
                                    const double xj = RECIP ? rdlane(sv, j) * rdlane(dg, j) : rdlane(sv, j) / rdlane(dg, j);
                                    sv              = dfma(-rrow[j], xj, sv);
                                    uint32_t tj = tid;
                                    asm volatile("" : "+v"(tj) : "v"(xj)); // (ties the lane test to the chain: sixty-four of them hoisted = 128 SGPRs, spilled)
                                    xfin = (tj == (uint32_t)j) ? xj : xfin;
                                }
                            });
                            sv = xfin;
                            if (tid < nb) xs[Fc + jb + tid] = sv;
                        }
                        __syncthreads();
                        SSTAMP(2)
                        for (uint32_t i = tid; i < jb; i += NT) // rows above the block: its columns, last first
                        {
                            double sv = xs[Fc + i];
                            uint32_t j = nb;
                            for (; j >= 64; j -= 64) // (a full block: its 64 entries of the row in ONE trip)
                            {
                                double w64[64];
#pragma unroll
                                for (uint32_t u = 0; u < 64; u++) w64[u] = W[F + i + (Fc + jb + j - 1 - u) * ld];
#pragma unroll
                                for (uint32_t u = 0; u < 64; u++) sv = dfma(-w64[u], xs[Fc + jb + j - 1 - u], sv);
                            }
                            for (; j >= 8; j -= 8)
                            {
                                double w8[8];
#pragma unroll
                                for (uint32_t u = 0; u < 8; u++) w8[u] = W[F + i + (Fc + jb + j - 1 - u) * ld];
#pragma unroll
                                for (uint32_t u = 0; u < 8; u++) sv = dfma(-w8[u], xs[Fc + jb + j - 1 - u], sv);
                            }
                            for (; j--;) sv = dfma(-W[F + i + (Fc + jb + j) * ld], xs[Fc + jb + j], sv);
                            xs[Fc + i] = sv;
                        }
                        __syncthreads();
                        SSTAMP(3)
                        if (jb == 0) break;
                    }
                }
                else
                {
                // the diagonal of R_k once into LDS; the column of the NEXT step is requested before the current step's division, so that a
                // step costs one division + one barrier instead of two dependent trips to the stored factor (same operations, same order)
                double *dgl = xs + n;
                for (uint32_t i = tid; i < rank; i += NT) dgl[i] = W[F + i + (Fc + i) * ld];
                double wcur = (rank >= 2 && tid + 1 < rank) ? W[F + tid + (Fc + rank - 1) * ld] : 0.0; // my row's entry of column rank-1
                __syncthreads();
                double xprev = 0.0;
                for (uint32_t j = rank; j--;)
                {
                    const double wnext = (j >= 1 && tid + 1 < j) ? W[F + tid + (Fc + j - 1) * ld] : 0.0;
                    if (j + 1 < rank && tid == 0) xs[Fc + j + 1] = xprev; // (nobody reads that entry any more in this loop)
                    const double xj = xs[Fc + j] / dgl[j];                 // every thread for itself: no hand-off
                    if (tid < j) xs[Fc + tid] = dfma(-wcur, xj, xs[Fc + tid]);
                    for (uint32_t i = tid + NT; i < j; i += NT) xs[Fc + i] = dfma(-W[F + i + (Fc + j) * ld], xj, xs[Fc + i]);
                    __syncthreads();
                    wcur  = wnext;
                    xprev = xj;
                }
                if (tid == 0) xs[Fc] = xprev;
                __syncthreads();
                }
                acc += rank;
            }
            // x = P x (lexlse.h:1044); scratch: the diagonal's place (free now), 16-byte aligned: xs = smem, an even number of doubles further on
            apply_permutation_by_following<NT>(a, b, perm, a.totalrank[b], xs, reinterpret_cast<uint32_t *>(xs + ((n + 1u) & ~1u)), tid);
            SSTAMP(4)
#ifdef LEXLS_SOLVE_STAMPS
            if (tid == 0)
                for (int i_ = 0; i_ < 5; i_++) a.lambda[(size_t)b * (n + cap) + 40 + i_] = (double)sst[i_];
#endif
        }

        // -----------------------------------------------------------------------------------------
        // get_v(): residuals through Q (lexlse.h:1560-1582)
        // -----------------------------------------------------------------------------------------
        template <int NT>
        __global__ __launch_bounds__(NT) void residual_kernel(LseArgs a)
        {
            extern __shared__ double smem[];
            const uint32_t b = blockIdx.x, tid = threadIdx.x;
            const uint32_t n = a.nVar, cap = a.cap, nObj = a.nObj;
            const double *W  = a.fac + (size_t)b * cap * (n + 1);
            const double *hh = a.hh + (size_t)b * cap;
            double *v        = smem;       // cap
            double *bcast    = smem + cap; // 1
            const uint32_t *dims = a.dims + (size_t)b * nObj;
            for (uint32_t i = tid; i < cap; i += NT) v[i] = 0.0;
            __syncthreads();
            uint32_t F = 0;
            for (uint32_t k = 0; k < nObj; k++)
            {
                const uint32_t dim = dims[k], rank = a.rank[(size_t)b * nObj + k], Fc = a.fcol[(size_t)b * nObj + k];
                for (uint32_t i = rank + tid; i < dim; i += NT) v[F + i] = -W[F + i + (size_t)n * cap];
                __syncthreads();
                apply_q_block<NT>(W, cap, hh, F, Fc, dim, rank, v + F, bcast, tid);
                F += dim;
            }
            __syncthreads();
            for (uint32_t i = tid; i < cap; i += NT) a.v[(size_t)b * cap + i] = v[i];
        }

        /// One coalesced pass over a problem's stored factor (cap x (nVar+1), column-major) into LDS, odd leading dimension so that
        /// "thread = row" and "thread = column" walks are conflict-free.  For kernels that walk the factor along dependent chains
        /// when latency, not occupancy, is what counts (few problems per CU; several levels per launch).
        template <int NT>
        __device__ __forceinline__ void stage_factor(const double *__restrict__ G, double *L, uint32_t cap, uint32_t ncol, uint32_t ldl, uint32_t tid)
        {
            constexpr uint32_t U = NT <= 64 ? 20 : 8; // loads in flight per thread (one wavefront staging for itself: the rounds are latency, not bandwidth)
            const uint32_t total = cap * ncol;
            for (uint32_t base = tid; base < total; base += NT * U)
            {
                double v[U];
#pragma unroll
                for (uint32_t u = 0; u < U; u++)
                {
                    const uint32_t e = base + u * NT;
                    v[u]             = e < total ? G[e] : 0.0;
                }
#pragma unroll
                for (uint32_t u = 0; u < U; u++)
                {
                    const uint32_t e = base + u * NT;
                    if (e < total)
                    {
                        const uint32_t j = e / cap;
                        L[(e - j * cap) + j * ldl] = v[u];
                    }
                }
            }
            __syncthreads();
        }

        // -----------------------------------------------------------------------------------------
        // ObjectiveSensitivity (lexlse.h:611-762) + findDescentDirection (:935-987)
        // -----------------------------------------------------------------------------------------
        struct SensState
        {
            double maxabs;
            uint32_t ctr;
            int obj, found;
        };

        /// scan one group of multipliers; sequential semantics of the reference (strict '<' keeps the first)
        __device__ void find_descent(uint8_t *types, const double *lambda, uint32_t count, double tolW, double tolC, int objTag, SensState *st)
        {
            for (uint32_t k = 0; k < count; k++)
            {
                const uint8_t t = types[k];
                if (t == CTR_ACTIVE_EQ || t == CORRECT_SIGN_OF_LAMBDA) continue;
                double al = lambda[k];
                if (t == CTR_ACTIVE_LB) al = -al;
                if (al > tolC)
                {
                    types[k] = CORRECT_SIGN_OF_LAMBDA;
                }
                else if (al < -tolW && al < st->maxabs)
                {
                    st->maxabs = al;
                    st->ctr    = k;
                    st->obj    = objTag;
                    st->found  = 1;
                }
            }
        }

        template <int NT, bool STAGE>
        __global__ __launch_bounds__(NT) void sensitivity_kernel(LseArgs a, const int32_t *obj_index, int32_t obj_all, double tolW, double tolC, int scan_up)
        {
            extern __shared__ double smem[];
            const uint32_t b = blockIdx.x, tid = threadIdx.x;
            const uint32_t n = a.nVar, cap = a.cap, nObj = a.nObj;
            const int32_t oi = obj_index ? obj_index[b] : obj_all;
            int32_t *sens    = a.sens + (size_t)b * 3;
            if (oi < 0 || (uint32_t)oi >= nObj)
            {
                if (tid == 0)
                {
                    sens[0]     = 0;
                    sens[1]     = -1;
                    sens[2]     = -2;
                    a.maxabs[b] = 0.0;
                }
                return;
            }
            const double *Wsrc  = a.fac + (size_t)b * cap * (n + 1);
            const double *hhsrc = a.hh + (size_t)b * cap;
            size_t ldsrc        = cap;
            uint8_t *types_l    = nullptr; // STAGE: LDS copy of the activation types (find_descent reads and marks one entry after the other)
            if (STAGE) // once per launch, shared by all the levels of a scan
            {
                // LDS: [LambdaFixed nVar | Lambda cap | rhs nVar | bcast | state | staged factor | Householder scalars]
                double *Wl  = smem + 2 * n + cap + 1 + (sizeof(SensState) + 7) / 8;
                double *hhl = Wl + (size_t)(cap | 1u) * (n + 1);
                for (uint32_t i = tid; i < cap; i += NT) hhl[i] = hhsrc[i]; // (every reflector of every level starts with its tau)
                types_l = reinterpret_cast<uint8_t *>(hhl + cap); // activation types: cap constraint rows, then nVar fixed variables
                for (uint32_t i = tid; i < cap; i += NT) types_l[i] = a.ctr_type[(size_t)b * cap + i];
                for (uint32_t i = tid; i < n; i += NT) types_l[cap + i] = a.fixed_type[(size_t)b * n + i];
                stage_factor<NT>(Wsrc, Wl, cap, n + 1, cap | 1u, tid);
                Wsrc  = Wl;
                hhsrc = hhl;
                ldsrc = cap | 1u;
            }
            // scan_up: what LexLSI's removal search does with one call per level (lexlsi.h:1121-1132) — levels oi, oi+1, ... until one
            // reports a wrong-sign multiplier or the last one is done — in ONE launch; the marks of a level are in place before the next
            for (uint32_t ObjIndex = (uint32_t)oi;; ObjIndex++)
            {
            const double *W  = Wsrc;
            const size_t ld  = ldsrc;
            const double *hh = hhsrc;
            const uint32_t *dims = a.dims + (size_t)b * nObj;
            const uint32_t *rk = a.rank + (size_t)b * nObj, *fc = a.fcol + (size_t)b * nObj;
            const uint32_t nf  = a.nfixed ? a.nfixed[b] : 0;
            uint8_t *ctr_type  = STAGE ? types_l : a.ctr_type + (size_t)b * cap;
            uint8_t *fix_type  = STAGE ? types_l + cap : a.fixed_type + (size_t)b * n;

            // LDS: [LambdaFixed nVar | Lambda cap | rhs nVar | bcast | state]
            double *LambdaFixed = smem;
            double *Lambda      = smem + n;
            double *rhs         = Lambda + cap;
            double *bcast       = rhs + n;
            SensState *st       = reinterpret_cast<SensState *>(bcast + 1);

            uint32_t nLambda = 0, Fobj = 0;
            for (uint32_t k = 0; k <= ObjIndex; k++)
            {
                if (k == ObjIndex) Fobj = nLambda;
                nLambda += dims[k];
            }
            for (uint32_t i = tid; i < n + cap + n; i += NT) smem[i] = 0.0;
            if (tid == 0)
            {
                st->maxabs = 0.0;
                st->ctr    = 0;
                st->obj    = -2;
                st->found  = 0;
            }
            __syncthreads();

            uint32_t F = Fobj, Fc = fc[ObjIndex], dim = dims[ObjIndex], rank = rk[ObjIndex];
            if (a.reg_type == 7) // lexlse.h:647-651, :688-690 ("WARNING: TESTING" there): multipliers of the REGULARIZED problem
            {
                double *mu        = a.reg_mu + (size_t)b * reg_mu_doubles(n, nObj, cap);
                const double *res = mu + 2 * (size_t)nObj * n;
                if (tid == 0) // initialize_rhs (:1921-1959; oracle: initialize_rhs): ordered chains over at most nVar entries
                {
                    const double *X      = mu + (size_t)ObjIndex * n;
                    double *c            = mu + (size_t)(nObj + ObjIndex) * n; // X_mu_rhs.col(ObjIndex)
                    const uint32_t *perm = a.perm + (size_t)b * n;
                    const uint32_t TotalRank = a.totalrank[b];
                    const double f           = a.reg_factor[(size_t)b * nObj + ObjIndex];
                    for (uint32_t i = 0; i < n; i++) c[i] = X[i];
                    for (uint32_t k = 0; k < TotalRank; k++) // P' * .
                    {
                        const uint32_t j = perm[k];
                        const double t   = c[k];
                        c[k]             = c[j];
                        c[j]             = t;
                    }
                    for (uint32_t i = 0; i < n; i++) c[i] *= -f * f;
                    const uint32_t last = Fc + rank;
                    uint32_t Fk = 0, Fp = 0, nRank = 0;
                    for (uint32_t k = 0; k <= ObjIndex; k++)
                    {
                        const uint32_t Fck = fc[k], rkk = rk[k];
                        if (k > 0)
                        {
                            const uint32_t Fcp = fc[k - 1], rp = rk[k - 1], remain = last - Fck;
                            for (uint32_t j = 0; j < remain; j++)
                            {
                                double s = c[Fck + j];
                                for (uint32_t i = 0; i < rp; i++) s = dfma(-W[Fp + i + (Fck + j) * ld], c[Fcp + i], s);
                                c[Fck + j] = s;
                            }
                        }
                        for (uint32_t j = 0; j < rkk; j++) // R_k' z = c
                        {
                            double s = c[Fck + j];
                            for (uint32_t i = 0; i < j; i++) s = dfma(-W[Fk + i + (Fck + j) * ld], c[Fck + i], s);
                            c[Fck + j] = s / W[Fk + j + (Fck + j) * ld];
                        }
                        if (k < ObjIndex) nRank += rkk;
                        Fp = Fk;
                        Fk += dims[k];
                    }
                    for (uint32_t i = 0; i < nRank + nf; i++) rhs[i] = c[i];
                }
                for (uint32_t i = tid; i < dim; i += NT) Lambda[F + i] = res[F + i];
                __syncthreads();
            }
            else
            {
            for (uint32_t i = rank + tid; i < dim; i += NT) Lambda[F + i] = -W[F + i + n * ld];
            __syncthreads();
            if (STAGE && NT == 64 && dim <= 64)
                apply_q_wave(W, ld, hh, F, Fc, dim, rank, Lambda + F, tid);
            else
                apply_q_block<NT>(W, ld, hh, F, Fc, dim, rank, Lambda + F, bcast, tid);
            }
            if (tid == 0) find_descent(ctr_type + F, Lambda + F, dim, tolW, tolC, (int)ObjIndex, st);

            if (ObjIndex > 0)
            {
                for (uint32_t c = tid; c < Fc; c += NT) // rhs.head(ColDim) -= L^T lambda (lexlse.h:706-707)
                {
                    rhs[c] -= chain_dot<8>(W + F + c * ld, Lambda + F, dim, 0.0);
                }
                __syncthreads();
                for (uint32_t k = ObjIndex; k--;)
                {
                    dim  = dims[k];
                    F    = F - dim;
                    Fc   = fc[k];
                    rank = rk[k];
                    for (uint32_t i = tid; i < rank; i += NT) Lambda[F + i] = rhs[Fc + i];
                    __syncthreads();
                    if (STAGE && NT == 64 && dim <= 64)
                        apply_q_wave(W, ld, hh, F, Fc, dim, rank, Lambda + F, tid);
                    else
                        apply_q_block<NT>(W, ld, hh, F, Fc, dim, rank, Lambda + F, bcast, tid);
                    for (uint32_t c = tid; c < Fc; c += NT)
                    {
                        rhs[c] -= chain_dot<8>(W + F + c * ld, Lambda + F, dim, 0.0);
                    }
                    if (tid == 0) find_descent(ctr_type + F, Lambda + F, dim, tolW, tolC, (int)k, st);
                    __syncthreads();
                }
            }

            if (nf > 0) // lexlse.h:742-758
            {
                __syncthreads();
                for (uint32_t c = tid; c < nf; c += NT)
                {
                    double s = 0.0;
                    for (uint32_t i = 0; i < nLambda; i++) s = dfma(W[i + c * ld], Lambda[i], s);
                    LambdaFixed[c] = -s;
                }
                __syncthreads();
                if (tid == 0) find_descent(fix_type, LambdaFixed, nf, tolW, tolC, -1, st);
            }
            __syncthreads();

            double *out = a.lambda + (size_t)b * (n + cap);
            for (uint32_t i = tid; i < n + cap; i += NT)
            {
                double val = 0.0;
                if (i < nf)
                    val = LambdaFixed[i];
                else if (i < nf + nLambda)
                    val = Lambda[i - nf];
                out[i] = val;
            }
            if (tid == 0)
            {
                sens[0]     = st->found;
                sens[1]     = st->found ? (int32_t)st->ctr : -1;
                sens[2]     = st->found ? st->obj : -2;
                a.maxabs[b] = st->maxabs;
            }
            const int found = st->found; // (LDS, written before the last barrier)
            if (!scan_up || found || ObjIndex + 1 >= nObj) break;
            __syncthreads(); // the next level re-initialises the LDS arrays; its find_descent sees this level's marks (same thread)
            }
            if (STAGE) // the CORRECT_SIGN_OF_LAMBDA marks go back to the arrays lexls_lse_get_ctr_type / _fixed_type read
            {
                __syncthreads();
                for (uint32_t i = tid; i < cap; i += NT) a.ctr_type[(size_t)b * cap + i] = types_l[i];
                for (uint32_t i = tid; i < n; i += NT) a.fixed_type[(size_t)b * n + i] = types_l[cap + i];
            }
        }

        // -----------------------------------------------------------------------------------------
        // solveLeastNorm_1(): Givens sweep (lexlse.h:1052-1131).  Per problem `scratch` holds
        // RT (nVar x nVar, ld = nVar) followed by the rotation table (c,s pairs, <= nVar^2/2 doubles).
        // -----------------------------------------------------------------------------------------
        __device__ __forceinline__ void make_givens(double p2, double q, double &c, double &s)
        {
            if (q == 0.0)
            {
                c = p2 < 0.0 ? -1.0 : 1.0;
                s = 0.0;
            }
            else if (p2 == 0.0)
            {
                c = 0.0;
                s = q < 0.0 ? 1.0 : -1.0;
            }
            else if (fabs(p2) > fabs(q))
            {
                const double t = q / p2;
                double u       = sqrt(dfma(t, t, 1.0));
                if (p2 < 0.0) u = -u;
                c = 1.0 / u;
                s = -t * c;
            }
            else
            {
                const double t = p2 / q;
                double u       = sqrt(dfma(t, t, 1.0));
                if (q < 0.0) u = -u;
                s = -1.0 / u;
                c = -t * s;
            }
        }

        template <int NT>
        __global__ __launch_bounds__(NT) void leastnorm_kernel(LseArgs a)
        {
            extern __shared__ double smem[];
            const uint32_t b = blockIdx.x, tid = threadIdx.x;
            const uint32_t n = a.nVar, cap = a.cap, nObj = a.nObj;
            const double *W  = a.fac + (size_t)b * cap * (n + 1);
            const size_t ld  = cap;
            const uint32_t *dims = a.dims + (size_t)b * nObj;
            const uint32_t *rk = a.rank + (size_t)b * nObj, *fc = a.fcol + (size_t)b * nObj;
            const uint32_t *perm = a.perm + (size_t)b * n;
            const uint32_t nf    = a.nfixed ? a.nfixed[b] : 0;

            uint32_t nVarRank = 0;
            for (uint32_t k = 0; k < nObj; k++) nVarRank += rk[k];
            const uint32_t nVarFree = n - (nVarRank + nf);
            const uint32_t ncol     = nVarRank + nVarFree;
            double *RT              = a.scratch + (size_t)b * 2 * n * n; // nVarRank x ncol, ld = n
            double *rot             = RT + (size_t)n * n;               // (c, s) pairs in sweep order
            const size_t lr         = n;
            double *rhs             = smem;         // n
            double *xs              = smem + n;     // n
            double *cs              = smem + 2 * n; // 2

            for (uint32_t e = tid; e < n * n; e += NT) RT[e] = 0.0;
            for (uint32_t i = tid; i < n; i += NT)
            {
                rhs[i] = 0.0;
                xs[i]  = (i < nf) ? a.fixed_val[(size_t)b * n + i] : 0.0;
            }
            __syncthreads();
            {
                uint32_t counter = 0, col_dim = ncol, F = 0;
                for (uint32_t k = 0; k < nObj; k++) // compact [R T | rhs] copy (lexlse.h:1081-1094)
                {
                    const uint32_t rank = rk[k], Fc = fc[k];
                    for (uint32_t e = tid; e < rank * col_dim; e += NT)
                    {
                        const uint32_t i = e % rank, j = e / rank;
                        if (j >= i) RT[counter + i + (counter + j) * lr] = W[F + i + (Fc + j) * ld];
                    }
                    for (uint32_t i = tid; i < rank; i += NT) rhs[counter + i] = W[F + i + n * ld];
                    counter += rank;
                    col_dim -= rank;
                    F += dims[k];
                }
            }
            __syncthreads();

            uint32_t nrot = 0;
            for (uint32_t i = 0; i < nVarFree; i++) // zero T against R from the right (lexlse.h:1099-1110)
            {
                for (uint32_t j = nVarRank; j--;)
                {
                    if (tid == 0)
                    {
                        double c, s;
                        make_givens(RT[j + j * lr], RT[j + (nVarRank + i) * lr], c, s);
                        cs[0]             = c;
                        cs[1]             = s;
                        rot[2 * nrot]     = c;
                        rot[2 * nrot + 1] = s;
                    }
                    __syncthreads();
                    const double c = cs[0], s = cs[1];
                    for (uint32_t r = tid; r <= j; r += NT) // applyOnTheRight on RT.topRows(j+1)
                    {
                        const double x1 = RT[r + j * lr], y1 = RT[r + (nVarRank + i) * lr];
                        RT[r + j * lr]              = dfma(c, x1, -(s * y1));
                        RT[r + (nVarRank + i) * lr] = dfma(c, y1, s * x1);
                    }
                    nrot++;
                    __syncthreads();
                }
            }

            for (uint32_t j = nVarRank; j--;) // R^-1 rhs, column-oriented (lexlse.h:1115)
            {
                if (tid == 0) rhs[j] = rhs[j] / RT[j + j * lr];
                __syncthreads();
                const double xj = rhs[j];
                for (uint32_t i = tid; i < j; i += NT) rhs[i] = dfma(-RT[i + j * lr], xj, rhs[i]);
                __syncthreads();
            }

            if (tid == 0)
            {
                for (uint32_t k = nrot; k--;) // replay on the RHS in reverse (lexlse.h:1121-1124)
                {
                    const uint32_t fi = k / nVarRank, jj = nVarRank - 1 - (k % nVarRank);
                    const uint32_t gi = jj, gj = nVarRank + fi;
                    const double c = rot[2 * k], s = rot[2 * k + 1];
                    const double x1 = rhs[gi], y1 = rhs[gj];
                    rhs[gi] = dfma(c, x1, s * y1);
                    rhs[gj] = dfma(c, y1, -(s * x1));
                }
                for (uint32_t i = 0; i < ncol; i++) xs[nf + i] = rhs[i]; // lexlse.h:1129
                for (uint32_t k = a.totalrank[b]; k--;)
                {
                    const uint32_t pk = perm[k];
                    const double t    = xs[k];
                    xs[k]             = xs[pk];
                    xs[pk]            = t;
                }
            }
            __syncthreads();
            for (uint32_t i = tid; i < n; i += NT) a.x[(size_t)b * n + i] = xs[i];
        }
        /// solveLeastNorm_2 (lexlse.h:1138-1213): least-norm solution through the normal equations of the free variables.
        /// One 64-lane workgroup per problem on the factor in HBM; [R T | rhs] and D live in the per-problem scratch.
        /// Arithmetic order as oracle/lexlse_oracle.h::solveLeastNorm_2 (bit-identical).
        template <int NT>
        __global__ __launch_bounds__(NT) void leastnorm2_kernel(LseArgs a)
        {
            extern __shared__ double smem[];
            const uint32_t b = blockIdx.x, tid = threadIdx.x;
            const uint32_t n = a.nVar, cap = a.cap, nObj = a.nObj;
            const double *W  = a.fac + (size_t)b * cap * (n + 1);
            const size_t ld  = cap;
            const uint32_t *dims = a.dims + (size_t)b * nObj;
            const uint32_t *rk = a.rank + (size_t)b * nObj, *fc = a.fcol + (size_t)b * nObj;
            const uint32_t *perm = a.perm + (size_t)b * n;
            const uint32_t nf    = a.nfixed ? a.nfixed[b] : 0;

            uint32_t nVarRank = 0;
            for (uint32_t k = 0; k < nObj; k++) nVarRank += rk[k];
            const uint32_t nVarFree = n - (nVarRank + nf);
            const uint32_t ncol     = nVarRank + nVarFree;
            const size_t lr         = nVarRank ? nVarRank : 1;
            double *RT              = a.scratch + (size_t)b * 2 * n * n; // nVarRank x (ncol + 1), column-major, ld = nVarRank
            double *D               = RT + lr * (ncol + 1);              // nVarFree x nVarFree, column-major (lower part used)
            const size_t ldd        = nVarFree ? nVarFree : 1;
            double *d               = smem;     // n
            double *xs              = smem + n; // n

            for (uint32_t e = tid; e < lr * (ncol + 1); e += NT) RT[e] = 0.0;
            for (uint32_t i = tid; i < n; i += NT)
            {
                d[i]  = 0.0;
                xs[i] = (i < nf) ? a.fixed_val[(size_t)b * n + i] : 0.0;
            }
            __syncthreads();
            {
                uint32_t counter = 0, col_dim = ncol, F = 0;
                for (uint32_t k = 0; k < nObj; k++) // compact [R T | rhs] copy (lexlse.h:1166-1177)
                {
                    const uint32_t rank = rk[k], Fc = fc[k];
                    for (uint32_t e = tid; e < rank * col_dim; e += NT)
                    {
                        const uint32_t i = e % rank, j = e / rank;
                        if (j >= i) RT[counter + i + (counter + j) * lr] = W[F + i + (Fc + j) * ld];
                    }
                    for (uint32_t i = tid; i < rank; i += NT) RT[counter + i + ncol * lr] = W[F + i + n * ld];
                    counter += rank;
                    col_dim -= rank;
                    F += dims[k];
                }
            }
            __syncthreads();
            // T <- R^-1 [T | rhs]: one column per lane, column-oriented back-substitution (lexlse.h:1180)
            for (uint32_t c = nVarRank + tid; c <= ncol; c += NT)
                for (uint32_t j = nVarRank; j--;)
                {
                    const double t = RT[j + c * lr] / RT[j + j * lr];
                    RT[j + c * lr] = t;
                    for (uint32_t i = 0; i < j; i++) RT[i + c * lr] = dfma(-RT[i + j * lr], t, RT[i + c * lr]);
                }
            __syncthreads();
            // D = I + T^T T (lower), d = T^T t_rhs (lexlse.h:1182-1188)
            for (uint32_t e = tid; e < nVarFree * nVarFree; e += NT)
            {
                const uint32_t i = e % nVarFree, j = e / nVarFree;
                if (i < j) continue;
                double acc = 0.0;
                for (uint32_t k = 0; k < nVarRank; k++) acc = dfma(RT[k + (nVarRank + i) * lr], RT[k + (nVarRank + j) * lr], acc);
                D[i + j * ldd] = (i == j) ? acc + 1.0 : acc;
            }
            for (uint32_t j = tid; j < nVarFree; j += NT)
            {
                double acc = 0.0;
                for (uint32_t k = 0; k < nVarRank; k++) acc = dfma(RT[k + (nVarRank + j) * lr], RT[k + ncol * lr], acc);
                d[j] = acc;
            }
            __syncthreads();
            for (uint32_t j = 0; j < nVarFree; j++) // Cholesky, left-looking by columns (lexlse.h:1190)
            {
                if (tid == 0)
                {
                    double sjj = D[j + j * ldd];
                    for (uint32_t k = 0; k < j; k++) sjj = dfma(-D[j + k * ldd], D[j + k * ldd], sjj);
                    D[j + j * ldd] = sqrt(sjj);
                }
                __syncthreads();
                const double ljj = D[j + j * ldd];
                for (uint32_t i = j + 1 + tid; i < nVarFree; i += NT)
                {
                    double v = D[i + j * ldd];
                    for (uint32_t k = 0; k < j; k++) v = dfma(-D[i + k * ldd], D[j + k * ldd], v);
                    D[i + j * ldd] = v / ljj;
                }
                __syncthreads();
            }
            for (uint32_t j = 0; j < nVarFree; j++) // L y = d
            {
                if (tid == 0) d[j] = d[j] / D[j + j * ldd];
                __syncthreads();
                const double yj = d[j];
                for (uint32_t i = j + 1 + tid; i < nVarFree; i += NT) d[i] = dfma(-D[i + j * ldd], yj, d[i]);
                __syncthreads();
            }
            for (uint32_t j = nVarFree; j--;) // L^T z = y
            {
                if (tid == 0) d[j] = d[j] / D[j + j * ldd];
                __syncthreads();
                const double zj = d[j];
                for (uint32_t i = tid; i < j; i += NT) d[i] = dfma(-D[j + i * ldd], zj, d[i]);
                __syncthreads();
            }
            for (uint32_t i = tid; i < nVarFree; i += NT) xs[nf + nVarRank + i] = d[i];
            __syncthreads();
            {
                uint32_t counter = 0, F = 0;
                for (uint32_t k = 0; k < nObj; k++) // x_rank = rhs - T_LOD x_free (lexlse.h:1193-1204)
                {
                    const uint32_t rank = rk[k];
                    for (uint32_t i = tid; i < rank; i += NT)
                    {
                        double acc = 0.0;
                        for (uint32_t c = 0; c < nVarFree; c++) acc = dfma(W[F + i + (size_t)(nVarRank + nf + c) * ld], xs[nf + nVarRank + c], acc);
                        xs[nf + counter + i] = W[F + i + n * ld] - acc;
                    }
                    counter += rank;
                    F += dims[k];
                }
            }
            __syncthreads();
            for (uint32_t j = nVarRank; j--;) // R^-1 on x.segment(nVarFixed, nVarRank) (lexlse.h:1205)
            {
                if (tid == 0) xs[nf + j] = xs[nf + j] / RT[j + j * lr];
                __syncthreads();
                const double xj = xs[nf + j];
                for (uint32_t i = tid; i < j; i += NT) xs[nf + i] = dfma(-RT[i + j * lr], xj, xs[nf + i]);
                __syncthreads();
            }
            if (tid == 0)
                for (uint32_t k = a.totalrank[b]; k--;) // x = P x
                {
                    const uint32_t pk = perm[k];
                    const double t    = xs[k];
                    xs[k]             = xs[pk];
                    xs[pk]            = t;
                }
            __syncthreads();
            for (uint32_t i = tid; i < n; i += NT) a.x[(size_t)b * n + i] = xs[i];
        }
        /// solveLeastNorm_3 (lexlse.h:1222-1277): least-norm solution from the null-space basis of the Tikhonov family,
        /// null_space(0:nVarRank, nVarFixed:) = [inv(R) | -inv(R) [T | rhs]].  Arithmetic order as the oracle's solveLeastNorm_3.
        template <int NT>
        __global__ __launch_bounds__(NT) void leastnorm3_kernel(LseArgs a)
        {
            extern __shared__ double smem[];
            const uint32_t b = blockIdx.x, tid = threadIdx.x;
            const uint32_t n = a.nVar, cap = a.cap, nObj = a.nObj;
            const double *W  = a.fac + (size_t)b * cap * (n + 1);
            const size_t ld  = cap;
            const double *NS = a.reg_scratch + (size_t)b * reg_scratch_doubles(n); // n x (n+1), ld = n
            const uint32_t *dims = a.dims + (size_t)b * nObj;
            const uint32_t *rk   = a.rank + (size_t)b * nObj;
            const uint32_t *perm = a.perm + (size_t)b * n;
            const uint32_t nf    = a.nfixed ? a.nfixed[b] : 0;
            uint32_t nVarRank    = 0;
            for (uint32_t k = 0; k < nObj; k++) nVarRank += rk[k];
            const uint32_t nVarFree = n - (nVarRank + nf);
            const uint32_t c0       = nf + nVarRank;
            double *D               = a.scratch + (size_t)b * 2 * n * n; // nVarFree x nVarFree (lower), column-major
            const size_t ldd        = nVarFree ? nVarFree : 1;
            double *d               = smem;     // n
            double *xs              = smem + n; // n
            double *out             = smem + 2 * n; // n

            for (uint32_t i = tid; i < n; i += NT) xs[i] = (i < nf) ? a.fixed_val[(size_t)b * n + i] : 0.0;
            for (uint32_t e = tid; e < nVarFree * nVarFree; e += NT)
            {
                const uint32_t i = e % nVarFree, j = e / nVarFree;
                if (i < j) continue;
                double acc = 0.0;
                for (uint32_t k = 0; k < nVarRank; k++) acc = dfma(NS[k + (size_t)(c0 + i) * n], NS[k + (size_t)(c0 + j) * n], acc);
                D[i + j * ldd] = (i == j) ? acc + 1.0 : acc;
            }
            for (uint32_t j = tid; j < nVarFree; j += NT)
            {
                double acc = 0.0;
                for (uint32_t k = 0; k < nVarRank; k++) acc = dfma(NS[k + (size_t)(c0 + j) * n], NS[k + (size_t)(c0 + nVarFree) * n], acc);
                d[j] = acc;
            }
            __syncthreads();
            for (uint32_t j = 0; j < nVarFree; j++) // Cholesky
            {
                if (tid == 0)
                {
                    double sjj = D[j + j * ldd];
                    for (uint32_t k = 0; k < j; k++) sjj = dfma(-D[j + k * ldd], D[j + k * ldd], sjj);
                    D[j + j * ldd] = sqrt(sjj);
                }
                __syncthreads();
                const double ljj = D[j + j * ldd];
                for (uint32_t i = j + 1 + tid; i < nVarFree; i += NT)
                {
                    double v = D[i + j * ldd];
                    for (uint32_t k = 0; k < j; k++) v = dfma(-D[i + k * ldd], D[j + k * ldd], v);
                    D[i + j * ldd] = v / ljj;
                }
                __syncthreads();
            }
            for (uint32_t j = 0; j < nVarFree; j++)
            {
                if (tid == 0) d[j] = d[j] / D[j + j * ldd];
                __syncthreads();
                const double yj = d[j];
                for (uint32_t i = j + 1 + tid; i < nVarFree; i += NT) d[i] = dfma(-D[i + j * ldd], yj, d[i]);
                __syncthreads();
            }
            for (uint32_t j = nVarFree; j--;)
            {
                if (tid == 0) d[j] = d[j] / D[j + j * ldd];
                __syncthreads();
                const double zj = d[j];
                for (uint32_t i = tid; i < j; i += NT) d[i] = dfma(-D[j + i * ldd], zj, d[i]);
                __syncthreads();
            }
            for (uint32_t i = tid; i < nVarFree; i += NT) xs[c0 + i] = d[i];
            __syncthreads();
            {
                uint32_t counter = 0, F = 0;
                for (uint32_t k = 0; k < nObj; k++)
                {
                    const uint32_t rank = rk[k];
                    for (uint32_t i = tid; i < rank; i += NT)
                    {
                        double acc = 0.0;
                        for (uint32_t c = 0; c < nVarFree; c++) acc = dfma(W[F + i + (size_t)(c0 + c) * ld], xs[c0 + c], acc);
                        xs[nf + counter + i] = W[F + i + n * ld] - acc;
                    }
                    counter += rank;
                    F += dims[k];
                }
            }
            __syncthreads();
            for (uint32_t i = tid; i < nVarRank; i += NT) // x_rank <- triu(inv(R)) * x_rank
            {
                double acc = 0.0;
                for (uint32_t j = i; j < nVarRank; j++) acc = dfma(NS[i + (size_t)(nf + j) * n], xs[nf + j], acc);
                out[i] = acc;
            }
            __syncthreads();
            for (uint32_t i = tid; i < nVarRank; i += NT) xs[nf + i] = out[i];
            __syncthreads();
            // x = P x (lexlse.h:1044).  Scratch: xs = smem + n here, so the 16-byte alignment is counted from smem — behind x (n doubles from
            // xs) rounded up to an even number of doubles FROM SMEM; it overlays `out`, which has been copied into xs above
            apply_permutation_by_following<NT>(a, b, perm, a.totalrank[b], xs, reinterpret_cast<uint32_t *>(smem + ((2u * n + 1u) & ~1u)), tid);
        }
    } // namespace

    // ---------------------------------------------------------------------------------------------
    // launchers
    // ---------------------------------------------------------------------------------------------
    namespace
    {
        template <typename K>
        hipError_t set_lds(K kernel, size_t bytes)
        {
            if (bytes <= 64 * 1024) return hipSuccess;
            return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        }

        size_t lqr_lds_bytes(const LseArgs &a, int NT, bool ldsmat)
        {
            size_t b = 8 * ((ldsmat ? (size_t)a.ldp * (a.nVar + 1) : 0) + 2 * (size_t)a.nVar + NT);
            b += sizeof(Shared) + 4 * ((size_t)NT + a.nVar + 3 * (size_t)a.nObj) + 16;
            return b;
        }

        template <int NT, bool LDSMAT>
        hipError_t launch_lqr_t(const LseArgs &a, bool write_factor, bool do_solve, hipStream_t s)
        {
            const size_t lds = lqr_lds_bytes(a, NT, LDSMAT);
            hipError_t e     = set_lds(lqr_generic_kernel<NT, LDSMAT>, lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((lqr_generic_kernel<NT, LDSMAT>), dim3(a.batch), dim3(NT), lds, s, a, write_factor ? 1 : 0, do_solve ? 1 : 0);
            return hipGetLastError();
        }
    } // namespace

    bool generic_fits_lds(const LseArgs &a0, uint32_t max_rows)
    {
        LseArgs a = a0;
        a.ldp     = odd_ld(max_rows);
        return lqr_lds_bytes(a, 1024, true) <= kMaxLdsBytes;
    }

    hipError_t launch_lqr_generic(LseArgs a, uint32_t max_rows, bool write_factor, bool do_solve, hipStream_t s, const char **variant)
    {
        a.ldp             = odd_ld(max_rows);
        const uint32_t w  = (a.nVar + 1 > max_rows) ? a.nVar + 1 : max_rows;
        const bool fits64 = lqr_lds_bytes(a, 64, true) <= kMaxLdsBytes;
        if (w <= 64 && fits64)
        {
            *variant = "lqr_generic<64,lds>";
            return launch_lqr_t<64, true>(a, write_factor, do_solve, s);
        }
        if (w <= 512 && lqr_lds_bytes(a, 256, true) <= kMaxLdsBytes)
        {
            *variant = "lqr_generic<256,lds>";
            return launch_lqr_t<256, true>(a, write_factor, do_solve, s);
        }
        if (lqr_lds_bytes(a, 1024, true) <= kMaxLdsBytes)
        {
            *variant = "lqr_generic<1024,lds>";
            return launch_lqr_t<1024, true>(a, write_factor, do_solve, s);
        }
        if (lqr_lds_bytes(a, 1024, false) > kMaxLdsBytes) return hipErrorInvalidValue; // nVar too large for the norm table
        *variant = "lqr_generic<1024,hbm>";
        return launch_lqr_t<1024, false>(a, true, do_solve, s);
    }

    hipError_t launch_solve_generic(const LseArgs &a, hipStream_t s, bool reciprocal_diagonal)
    {
        const size_t lds = 16 * (size_t)a.nVar + 16 + (a.nVar + 1 <= 64 ? 0 : 8 * 64 * 65); // x by position + the diagonal of the level being solved (+ the 64 x 64 block of the blocked form)
        if (a.nVar + 1 <= 64)
        {
            hipError_t e = set_lds(solve_generic_kernel<64>, lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((solve_generic_kernel<64>), dim3(a.batch), dim3(64), lds, s, a);
        }
        else
        {
            hipError_t e = reciprocal_diagonal ? set_lds(solve_generic_kernel<256, true>, lds) : set_lds(solve_generic_kernel<256>, lds);
            if (e != hipSuccess) return e;
            if (reciprocal_diagonal)
                hipLaunchKernelGGL((solve_generic_kernel<256, true>), dim3(a.batch), dim3(256), lds, s, a);
            else
                hipLaunchKernelGGL((solve_generic_kernel<256>), dim3(a.batch), dim3(256), lds, s, a);
        }
        return hipGetLastError();
    }

    namespace
    {
        /// LOD <- rows of the resident constraint data (Objective::formLexLSE, objective.h:434-494, done on the device): one
        /// workgroup per problem; consecutive threads write consecutive rows of one LOD column (coalesced stores)
        __global__ __launch_bounds__(256) void gather_rows_kernel(LseArgs a, const double *cdata, uint64_t per_problem, const uint32_t *row_src,
                                                                  const uint32_t *row_ld, double *dst)
        {
            const uint32_t b = blockIdx.x;
            if (a.skip && a.skip[b]) return;
            const uint32_t n = a.nVar, cap = a.cap;
            const double *src   = cdata + (size_t)b * per_problem;
            double *out         = dst + (size_t)b * cap * (n + 1);
            const uint32_t *rs  = row_src + (size_t)b * cap;
            const uint32_t *rl  = row_ld + (size_t)b * cap;
            for (uint32_t idx = threadIdx.x; idx < cap * (n + 1); idx += blockDim.x)
            {
                const uint32_t r = idx % cap, j = idx / cap;
                const uint32_t l = rl[r];
                if (l == 0) continue;
                const uint32_t ld = l & 0x7fffffffu, ub = l >> 31;
                out[idx] = src[rs[r] + (size_t)(j < n ? j : n + ub) * ld];
            }
        }

    } // namespace

    hipError_t launch_gather_rows(const LseArgs &a, const double *d_cdata, uint64_t per_problem, const uint32_t *d_row_src, const uint32_t *d_row_ld,
                                  double *d_dst, hipStream_t s)
    {
        hipLaunchKernelGGL(gather_rows_kernel, dim3(a.batch), dim3(256), 0, s, a, d_cdata, per_problem, d_row_src, d_row_ld, d_dst);
        return hipGetLastError();
    }

    hipError_t launch_residual(const LseArgs &a, hipStream_t s)
    {
        const size_t lds = 8 * ((size_t)a.cap + 2);
        if (lds > kMaxLdsBytes) return hipErrorInvalidValue;
        hipError_t e = set_lds(residual_kernel<64>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((residual_kernel<64>), dim3(a.batch), dim3(64), lds, s, a);
        return hipGetLastError();
    }

    namespace
    {
        int cus_of_current_device() // (a process may drive several)
        {
            static int cus_of[64] = {0};
            int dev = 0;
            if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64)
            {
                if (!cus_of[dev] && (hipDeviceGetAttribute(&cus_of[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus_of[dev] <= 0)) cus_of[dev] = 256;
                return cus_of[dev];
            }
            return 256;
        }
    } // namespace

    /// the single-sweep form (sensitivity_sweep_kernel): level dims <= 16, at most 8 objectives per sweep, one wavefront per problem with
    /// the factor staged in LDS — what a lock-step LSI stage asks for.  max_level_dim comes from the caller (0 = unknown: not taken)
    bool sensitivity_sweep_serves(const LseArgs &a, uint32_t sweep_level_dim_hint)
    {
        return a.reg_type != 7 && sweep_level_dim_hint > 0 && sweep_level_dim_hint <= (uint32_t)SWEEP_MD && a.nObj <= 8 && a.nVar <= 64 && sweep_lds_bytes(a) <= 64 * 1024 &&
               a.batch <= 4u * (uint32_t)cus_of_current_device() && !std::getenv("LEXLS_SENS_NO_SWEEP");
    }

    hipError_t launch_sensitivity(const LseArgs &a, const int32_t *d_obj_index, int32_t obj_all, double tolW, double tolC, hipStream_t s, bool scan_up, uint32_t sweep_level_dim_hint)
    {
        const size_t lds = 8 * (2 * (size_t)a.nVar + a.cap + 2) + sizeof(SensState) + 16;
        if (lds > kMaxLdsBytes) return hipErrorInvalidValue;
        // Factor staged into LDS when latency is what counts: few problems per CU (a lock-step LSI stage) — with thousands of problems
        // the 20 KB per workgroup would cost the wavefronts in flight that hide the chains instead (4096 problems: 0.090 -> 0.126 ms)
        const size_t lds_staged = lds + 8 * ((size_t)(a.cap | 1u) * (a.nVar + 1) + a.cap) + (((size_t)a.cap + a.nVar + 15) & ~(size_t)15);
        const int cus           = cus_of_current_device();
        const size_t lds_sweep  = sweep_lds_bytes(a);
        if (sensitivity_sweep_serves(a, sweep_level_dim_hint))
        {
            if (sweep_level_dim_hint <= 12)
                hipLaunchKernelGGL(sensitivity_sweep_kernel<12>, dim3(a.batch), dim3(64), lds_sweep, s, a, d_obj_index, obj_all, tolW, tolC, scan_up ? 1 : 0);
            else
                hipLaunchKernelGGL(sensitivity_sweep_kernel<SWEEP_MD>, dim3(a.batch), dim3(64), lds_sweep, s, a, d_obj_index, obj_all, tolW, tolC, scan_up ? 1 : 0);
            return hipGetLastError();
        }
        // (up to a problem per CU the whole LDS is there for the taking: a single mid-size problem — configs[0] — walks its chains at LDS
        // latency instead of through L2)
        if ((lds_staged <= 40 * 1024 && a.batch <= 4u * (uint32_t)cus) || (lds_staged <= kMaxLdsBytes && a.batch <= (uint32_t)cus))
        {
            if (lds_staged > 64 * 1024)
            {
                hipError_t e = set_lds(sensitivity_kernel<64, true>, lds_staged);
                if (e != hipSuccess) return e;
            }
            hipLaunchKernelGGL((sensitivity_kernel<64, true>), dim3(a.batch), dim3(64), lds_staged, s, a, d_obj_index, obj_all, tolW, tolC, scan_up ? 1 : 0);
            return hipGetLastError();
        }
        hipError_t e = set_lds(sensitivity_kernel<64, false>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((sensitivity_kernel<64, false>), dim3(a.batch), dim3(64), lds, s, a, d_obj_index, obj_all, tolW, tolC, scan_up ? 1 : 0);
        return hipGetLastError();
    }

    hipError_t launch_leastnorm(const LseArgs &a, hipStream_t s)
    {
        const size_t lds = 8 * (2 * (size_t)a.nVar + 4);
        if (lds > kMaxLdsBytes) return hipErrorInvalidValue;
        hipError_t e = set_lds(leastnorm_kernel<64>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((leastnorm_kernel<64>), dim3(a.batch), dim3(64), lds, s, a);
        return hipGetLastError();
    }

    hipError_t launch_leastnorm2(const LseArgs &a, hipStream_t s)
    {
        const size_t lds = 8 * (2 * (size_t)a.nVar + 4);
        if (lds > kMaxLdsBytes) return hipErrorInvalidValue;
        hipError_t e = set_lds(leastnorm2_kernel<64>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((leastnorm2_kernel<64>), dim3(a.batch), dim3(64), lds, s, a);
        return hipGetLastError();
    }

    hipError_t launch_leastnorm3(const LseArgs &a, hipStream_t s)
    {
        const size_t lds = 8 * (3 * (size_t)a.nVar + 4);
        if (lds > kMaxLdsBytes) return hipErrorInvalidValue;
        hipError_t e = set_lds(leastnorm3_kernel<64>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((leastnorm3_kernel<64>), dim3(a.batch), dim3(64), lds, s, a);
        return hipGetLastError();
    }
} // namespace lexls
