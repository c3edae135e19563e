// LEFT-LOOKING one-wavefront-per-problem lexicographic-QR kernel for IK-sized problems.
//
// Same results, bit for bit, as lqr_small_impl.h / lqr_generic.hip (arithmetic contract of oracle/lexlse_oracle.h) — what
// changes is WHEN the Gauss elimination of a level's rows happens.  The reference (lexlse.h:431-471) and the register-
// resident wave kernel eliminate ALL lower rows right after a level is factorised, so every row of the problem has to stay
// in registers (82 VGPRs for the 41 columns, 2 waves/SIMD).  Here a level's rows are only loaded when the level is reached
// and then absorb the eliminations of the earlier levels one after the other — each row still sees exactly the same
// sequence of fma's, in the same order, as in the right-looking form, because a row's update by level q depends only on
// that row and on [R_q T_q].  Live state per wave is therefore ONE level block:
//
//   hh[r]  (MD doubles / lane)  COLUMN-PER-LANE: lane j holds physical column j (lane n = RHS) of the level's <= MD rows.
//          Norms, Householder dots and the elimination chains are lane-local ordered fma chains; columns never move
//          (`pos` = position of physical column j in the reference's permuted order, lexlse.h:222-232).
//   IMG    (LDS) compact [R_q T_q | rhs_q] of the finished levels, row-major by pivot row: lane j reads ITS column of
//          level q (conflict-free), the back-substitution reads rows.  <= 1064 doubles for n = 40, level dims <= 12.
//
// ~10 KB of LDS and <= 128 VGPRs per wave -> 4 waves per SIMD (the register-resident kernel: 2), and the level blocks
// stream in when their level starts (an L2 warm-up by extra touch loads was measured: 4 % slower).
// Fixed variables are not handled here (the dispatcher keeps such batches on the register-resident kernel).
#pragma once
#include "lqr_wave_common.h"

namespace lexls
{
    namespace
    {
        /// orders this wave's LDS accesses for the compiler only (one wave per workgroup: the LDS pipeline keeps a wave's accesses in
        /// issue order); unlike __syncthreads() it does not wait for outstanding global loads / stores
        /// pins MD doubles at this point of the instruction stream (together with a memory clobber): the arithmetic that produces
        /// them cannot sink below, the LDS loads that follow cannot rise above — keeps the scheduler from issuing every load of an
        /// unrolled chain first and then spilling what it loaded
        template <int MD>
        __device__ __forceinline__ void pin_values(double (&v)[MD])
        {
            static_assert(MD == 12, "operand list below is written for 12 values");
            asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11])::"memory");
        }

        __device__ __forceinline__ void wave_lds_fence()
        {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            asm volatile("" ::: "memory"); // the scoped fence alone does not stop LLVM from hoisting plain LDS loads across it
        }

#ifndef LEXLS_LWAVE_OCC
#define LEXLS_LWAVE_OCC 4
#endif
        template <int NC, int MD, bool EXACT, bool WF>
// waves (= problems) per workgroup: the waves never synchronise with each other — fewer, fatter workgroups only shorten the dispatch ramp
#ifndef LEXLS_LWAVE_WPB
#define LEXLS_LWAVE_WPB 4
#endif
        __global__ __launch_bounds__(64 * LEXLS_LWAVE_WPB, LEXLS_LWAVE_OCC) void lqr_lwave_kernel(LseArgs a, uint32_t img_doubles, uint32_t lds_doubles_per_wave)
        {
            constexpr bool write_factor = WF;
            extern __shared__ double smem[];
            const int lane       = threadIdx.x & 63;
            const uint32_t wib   = threadIdx.x >> 6; // wave in block: owns its own slice of the dynamic LDS
            const uint32_t b     = blockIdx.x * LEXLS_LWAVE_WPB + wib;
            if (b >= a.batch) return; // no barrier anywhere in this kernel
            const int n          = EXACT ? NC - 1 : (int)a.nVar;
            const int cap        = (int)a.cap;
            const int nObj       = (int)a.nObj;
            const size_t pstride = (size_t)cap * (n + 1);
            if (a.skip && a.skip[b]) return; // uniform per wave

            // ---- LDS carve-up (see launch_lwave_t for the byte count) ----
            double *IMG      = smem + (size_t)wib * lds_doubles_per_wave;  // img_doubles: compact images of the levels
            double *xs       = IMG + img_doubles;                          // NC+1: solution by position (back-substitution only)
            double *EX       = xs;                                         // 2*16 : lane exchange while the levels are processed (aliases xs)
            double *EB       = xs + 16;
            double *idg_s    = xs + (NC + 1);                              // NC+1: 1/R_cc by pivot position c
            double *TB       = idg_s + (NC + 1);                           // 6*MD: half a multiplier block while a level is eliminated
            uint32_t *meta   = reinterpret_cast<uint32_t *>(TB + 6 * MD);  // 4*nObj: first column, rank, image base, image width
            uint16_t *offs   = reinterpret_cast<uint16_t *>(meta + 4 * nObj); // 64: image column of the c-th solved position
            uint8_t *pivl_s  = reinterpret_cast<uint8_t *>(offs + 64);     // 64: lane (physical column) of the pivot at position c
            uint8_t *perm_s  = pivl_s + 64;                                // 64: column_permutations
            uint8_t *phys_s  = perm_s + 64;                                // 64: physical column at each final position
            STAMP_DECL

            const uint32_t *dims = a.dims + (size_t)b * nObj;
            const double *in     = a.in + b * pstride;
            double *out          = a.fac + b * pstride;
            double *hhs          = a.hh + (size_t)b * cap;
            for (int i = lane; i < cap; i += 64) hhs[i] = 0.0; // initialize(), lexlse.h:1683
            perm_s[lane] = (uint8_t)lane;

            int pos       = (lane < n) ? lane : (lane == n ? n : 0x3fffffff);
            unsigned long long slots = 0ull; // byte k: position of this lane's column when level k was stored (nObj <= 8)
            int ColIndex  = 0;
            int TotalRank = 0;
            uint32_t imgp = 0; // bump pointer into IMG
            bool exhausted = false;
            int F          = 0;
            unsigned parked_levels = 0u; // (factor output) bit k: level k was written by physical column and still has to be re-ordered

            // column-per-lane load of one level block; 16-byte loads when the block starts on an even element
            auto load_level = [&](double (&dst)[MD], int Frow, int dim) {
                const double *src = in + Frow + (size_t)(lane <= n ? lane : 0) * cap;
                if (dim == MD && ((cap | Frow) & 1) == 0)
                {
                    const double2 *s2 = reinterpret_cast<const double2 *>(src);
#pragma unroll
                    for (int r = 0; r < MD / 2; r++)
                    {
                        const double2 v = s2[r];
                        dst[2 * r]      = v.x;
                        dst[2 * r + 1]  = v.y;
                    }
                }
                else
                {
#pragma unroll
                    for (int r = 0; r < MD; r++) dst[r] = (r < dim) ? src[r] : 0.0;
                }
                if (lane > n)
                {
#pragma unroll
                    for (int r = 0; r < MD; r++) dst[r] = 0.0;
                }
            };

            STAMP(0)

            for (int k = 0; k < nObj; k++)
            {
                const int dim_rt = (int)dims[k];
                const int Fc     = ColIndex;
                int rank         = 0;
                const bool work  = dim_rt > 0 && (!exhausted || write_factor);

                double hh[MD];
                if (work)
                    load_level(hh, F, dim_rt);
                else
                {
#pragma unroll
                    for (int r = 0; r < MD; r++) hh[r] = 0.0;
                }
                STAMP(1)

                // =====================================================================================
                // Gauss elimination of this level's rows by every finished level q < k (lexlse.h:431-471, left-looking):
                // for each pivot p of level q:  L_p = A[:, piv] / R_pp  (a product with the stored reciprocal), then every column
                // behind that pivot absorbs -L_p * R_q[p, column].  The multiplier column lives in the pivot's lane; it is
                // handed to the other lanes through SGPRs.
                // =====================================================================================
                auto eliminate = [&](auto full_c, int q) {
                    constexpr bool FULL = decltype(full_c)::value; // dim == MD and rank_q == MD: static instruction stream
                    constexpr int HB    = MD / 2;                  // pivots per half block
                    const int dim       = FULL ? MD : dim_rt;
                    const int rq        = FULL ? MD : uni((int)meta[4 * q + 1]);
                    const int Fcq       = uni((int)meta[4 * q + 0]);
                    const double *imgq  = IMG + uni((int)meta[4 * q + 2]);
                    const int wq        = n + 1 - Fcq;
                    const int myslot    = (int)((slots >> (8 * q)) & 0xffull) - Fcq;
                    const bool mine     = lane <= n && myslot >= 0; // this column was still behind level q's first pivot
                    const int myp       = (lane < n) ? pos - Fcq : -1; // 0 <= myp < rq: this lane holds the column of pivot myp of level q
                    const double *ucol  = imgq + (mine ? myslot : 0); // this lane's column of [R_q T_q | rhs_q]: entry p at ucol[p * wq]

                    // (1) the dim x rq block under level q's pivots, transposed: lane r gets row r (two half blocks through TB)
                    double arow[MD];
#pragma unroll
                    for (int h = 0; h < 2; h++)
                    {
                        if (myp >= h * HB && myp < (h + 1) * HB && myp < rq)
                        {
                            double *dst = TB + (myp - h * HB) * MD;
#pragma unroll
                            for (int r = 0; r < MD; r++) dst[r] = hh[r];
                        }
                        wave_lds_fence();
#pragma unroll
                        for (int p = 0; p < HB; p++) arow[h * HB + p] = TB[p * MD + (lane < MD ? lane : 0)];
                        wave_lds_fence();
                    }
                    // (2) L <- A R^-1 inside the lane that holds the row (lexlse.h:441-446): p ascending, every later column absorbs
                    //     L_p at once; R[p][p'] and 1/R_pp are wave-uniform LDS reads.  The pin after every step keeps the scheduler from
                    //     issuing all 66 reads first (it then spills what it loaded); fetching row p+1 by hand while row p is in use
                    //     was measured: more registers, no gain at 4 waves/SIMD
#pragma unroll
                    for (int p = 0; p < MD; p++)
                    {
                        if (!(FULL || p < rq)) continue;
                        arow[p] = arow[p] * idg_s[Fcq + p];
#pragma unroll
                        for (int p2 = 0; p2 < MD; p2++)
                            if (p2 > p && (FULL || p2 < rq)) arow[p2] = dfma(-arow[p], imgq[p * wq + p2], arow[p2]);
                        pin_values<MD>(arow); // one row of R in registers at a time
                    }
                    // (3) hand the multipliers to every lane, half a block at a time, and update the columns behind level q's pivots
                    //     (lexlse.h:448-471): per row an ordered chain over p, exactly as in the right-looking form
                    const bool behind = lane <= n && ((lane == n) || pos >= Fcq + rq);
#pragma unroll
                    for (int h = 0; h < 2; h++)
                    {
                        if (!(FULL || h * HB < rq)) continue;
                        if (lane < MD)
                        {
#pragma unroll
                            for (int p = 0; p < HB; p++) TB[p * MD + lane] = arow[h * HB + p];
                        }
                        wave_lds_fence();
#pragma unroll
                        for (int p = 0; p < HB; p++)
                        {
                            const int pa = h * HB + p;
                            if (!(FULL || pa < rq)) continue;
                            double Lb[MD];
#pragma unroll
                            for (int r = 0; r < MD; r++) Lb[r] = TB[p * MD + r]; // uniform address: broadcast read
                            if (behind)
                            {
                                const double up = ucol[pa * wq];
#pragma unroll
                                for (int r = 0; r < MD; r++)
                                    if (FULL || r < dim) hh[r] = dfma(-Lb[r], up, hh[r]);
                            }
                            if (write_factor && myp == pa)
                            {
#pragma unroll
                                for (int r = 0; r < MD; r++)
                                    if (FULL || r < dim) hh[r] = Lb[r];
                            }
                            pin_values<MD>(hh); // one multiplier column in registers at a time
                        }
                    }
                };
                if (work)
                {
                    for (int q = 0; q < k; q++)
                    {
                        const int rq = uni((int)meta[4 * q + 1]);
                        if (rq == 0) continue;
                        if (rq == MD && dim_rt == MD)
                            eliminate(std::true_type{}, q);
                        else
                            eliminate(std::false_type{}, q);
                    }
                }
                STAMP(7)

                // =====================================================================================
                // Householder QR with column pivoting of the level (lexlse.h:182-268)
                // =====================================================================================
                auto factor_level = [&](auto full_c) {
                    constexpr bool FULL = decltype(full_c)::value;
                    const int dim       = FULL ? MD : dim_rt;

                    // initial squared norms of the level's columns (lexlse.h:193-196); rows >= dim are zero: fma(0,0,s) == s
                    double nrm = 0.0;
#pragma unroll
                    for (int r = 0; r < MD; r++) nrm = dfma(hh[r], hh[r], nrm);

                    double mytau = 0.0; // lane r: tau of the level's row r (0 where no reflector was made, lexlse.h:239,1683)
                    bool go      = true; // wave-uniform: false once the level hit its rank / the columns ran out
#pragma unroll
                    for (int counter = 0; counter < MD; counter++)
                    {
                        if (!(go && counter < dim)) continue;
                        const int R   = dim - counter; // compile-time when FULL

                        // -- fresh norm and Householder tail norm of EVERY column (lexlse.h:210-211, :241): only the pivot's are used, but
                        //    the chains do not depend on the pivot search, so they fill its DPP / readlane latencies --
                        double fr = 0.0, tl = 0.0;
#pragma unroll
                        for (int r = 0; r < MD; r++)
                        {
                            if (r >= counter) fr = dfma(hh[r], hh[r], fr);
                            if (r > counter) tl = dfma(hh[r], hh[r], tl);
                        }
                        // -- pivot: first maximum (by position) of the down-dated norms (lexlse.h:205-206) --
                        const bool cand      = (lane < n) && (pos >= ColIndex);
                        const double key     = cand ? nrm : -INFINITY;
                        const double maxv    = wave_max(key);
                        unsigned long long m = __ballot(cand && key == maxv);
                        int pl               = (int)__builtin_ctzll(m);
                        if (__builtin_popcountll(m) > 1)
                        {
                            int bestpos = 0x7fffffff;
                            while (m)
                            {
                                const int l = (int)__builtin_ctzll(m);
                                m &= m - 1;
                                const int p2 = __builtin_amdgcn_readlane(pos, l);
                                if (p2 < bestpos)
                                {
                                    bestpos = p2;
                                    pl      = l;
                                }
                            }
                        }
                        pl = uni(pl);
                        STAMP(2)

                        const double fresh = rdlane(fr, pl);
                        if (lane == pl) nrm = fresh;
                        if (fresh < a.tol) // rank test on the squared norm (lexlse.h:214)
                        {
                            go = false;
                            continue;
                        }
                        STAMP(3)

                        // -- column "swap": update the position map (lexlse.h:222-232) --
                        const int ppos = __builtin_amdgcn_readlane(pos, pl);
                        if (lane == 0) perm_s[ColIndex] = (uint8_t)ppos;
                        {
                            const unsigned long long mc = __ballot(lane < n && pos == ColIndex);
                            const int lc                = (int)__builtin_ctzll(mc);
                            if (lane == lc) pos = ppos;
                            if (lane == pl) pos = ColIndex;
                        }

                        const double c0 = rdlane(hh[counter], pl);
                        if (R > 1)
                        {
                            const double tailSq   = rdlane(tl, pl);
                            const bool degenerate = tailSq <= DBL_MIN;
                            double beta           = sqrt(dfma(c0, c0, tailSq));
                            if (c0 >= 0.0) beta = -beta;
                            const double diag = degenerate ? c0 : beta;
                            const double den  = c0 - beta;

                            // spread the pivot column over the lanes: lane r gets v_r for the rows r below the pivot row, so
                            // that tau, the R-1 essentials and 1/R_jj cost ONE division sequence
                            if (lane == pl)
                            {
#pragma unroll
                                for (int r = 0; r < MD; r++)
                                    if (r > counter && (FULL || r < dim)) EX[r] = hh[r];
                            }
                            wave_lds_fence(); // same wave, LDS is in order: no barrier, no wait for outstanding global traffic
                            const double spread = EX[lane & 15];
                            const bool ess_lane = lane > counter && lane < dim;
                            double num          = ess_lane ? spread : 1.0;
                            double dnm          = ess_lane ? den : diag; // lanes without a role compute 1/diag (lane 63 is read)
                            if (lane == 0)
                            {
                                num = beta - c0;
                                dnm = beta;
                            }
                            const double quo = num / dnm; // tau | essential part | 1/R_jj
                            if (lane == 63) idg_s[ColIndex] = quo;
                            STAMP(4)
                            // wave-uniform tau and essentials (SGPR pairs); zero beyond the level's rows and when H is the identity
                            const double quo_e = (degenerate || !(ess_lane || lane == 0)) ? 0.0 : quo;
                            if (lane < 16) EB[lane] = quo_e;
                            wave_lds_fence(); // same wave, LDS is in order: no barrier, no wait for outstanding global traffic
                            double e[MD]; // e[r] = essential entry of row r (absolute row inside the level); wave-uniform values
#pragma unroll
                            for (int r = 0; r < MD; r++) e[r] = (r == 0 || r > counter) ? EB[r] : 0.0;
                            const double tau = e[0];

                            if constexpr (write_factor)
                            {
                                // the pivot column now holds beta and the essential part (zeros if degenerate)
                                if (lane == pl)
                                {
                                    hh[counter] = diag;
#pragma unroll
                                    for (int r = 0; r < MD; r++)
                                        if (r > counter) hh[r] = e[r];
                                }
                                // apply H to the trailing columns and the RHS (lexlse.h:243-246); zero essentials are no-ops
                                const bool trailing = ((lane < n) && (pos > ColIndex)) || (lane == n);
                                if (tau != 0.0 && trailing)
                                {
                                    double tmp = 0.0;
#pragma unroll
                                    for (int r = 0; r < MD; r++)
                                        if (r > counter) tmp = dfma(e[r], hh[r], tmp);
                                    tmp += hh[counter];
                                    hh[counter] = dfma(-tau, tmp, hh[counter]);
                                    const double ntau = -tau;
#pragma unroll
                                    for (int r = 0; r < MD; r++)
                                        if (r > counter) hh[r] = dfma(e[r] * ntau, tmp, hh[r]);
                                }
                            }
                            else
                            {
                                // x-only: below row `counter` the columns that are NOT trailing hold multipliers and essential parts, which
                                // are only ever read back from the factor — so H is applied to every lane (no divergent region, no
                                // register copies around it) and the pivot's diagonal entry is put in place afterwards
                                if (uni(tau != 0.0 ? 1 : 0))
                                {
                                    double tmp = 0.0;
#pragma unroll
                                    for (int r = 0; r < MD; r++)
                                        if (r > counter) tmp = dfma(e[r], hh[r], tmp);
                                    tmp += hh[counter];
                                    hh[counter] = dfma(-tau, tmp, hh[counter]);
                                    const double ntau = -tau;
#pragma unroll
                                    for (int r = 0; r < MD; r++)
                                        if (r > counter) hh[r] = dfma(e[r] * ntau, tmp, hh[r]);
                                }
                                if (lane == pl) hh[counter] = diag;
                            }
                            if (lane == counter) mytau = tau; // hh_scalars of this level leave in one store
                        }
                        else
                        {
                            if (lane == 63) idg_s[ColIndex] = 1.0 / c0;
                        }
                        if (lane == 0) pivl_s[ColIndex] = (uint8_t)pl;

                        ColIndex++;
                        rank++;
                        if (ColIndex == n)
                        {
                            exhausted = true;
                            go        = false;
                        }
                        else if ((lane < n) && (pos >= ColIndex))
                        {
                            nrm = dfma(-hh[counter], hh[counter], nrm); // lexlse.h:262-266
                        }
                        STAMP(5)
                    }
                    if (lane < dim) hhs[F + lane] = mytau;
                };
                if (work && !exhausted)
                {
                    if (dim_rt == MD)
                        factor_level(std::true_type{});
                    else
                        factor_level(std::false_type{});
                }
                const int dim = dim_rt;

                const int w = n + 1 - Fc;
                if (lane == 0)
                {
                    meta[4 * k + 0] = (uint32_t)Fc;
                    meta[4 * k + 1] = (uint32_t)rank;
                    meta[4 * k + 2] = imgp;
                    meta[4 * k + 3] = (uint32_t)w;
                }
                TotalRank += rank;

                // ---- the level's final rows [R_k T_k | rhs_k]: compact image for later eliminations and the back-substitution ----
                const int slot = (lane < n) ? pos : n;
                if (rank > 0)
                {
                    double *img = IMG + imgp;
                    if (lane <= n && slot >= Fc)
                    {
#pragma unroll
                        for (int r = 0; r < MD; r++)
                            if (r < rank) img[r * w + (slot - Fc)] = hh[r];
                    }
                    imgp += (uint32_t)(w * rank);
                }
                slots |= (unsigned long long)((lane < n) ? pos : n) << (8 * k);
                if (write_factor && dim > 0)
                {
                    // Factor rows of this level.  A column that has been pivoted (and the RHS) already sits at its final position and
                    // goes there directly.  The still-free columns of a full-rank level hold nothing but T entries, which are in the LDS
                    // image: they are written from there once their final positions are known (end of the kernel).  Only a
                    // rank-deficient level (free columns then also carry rows below the rank) is parked by PHYSICAL column and
                    // re-ordered at the end.
                    const bool parked = rank != dim && ColIndex < n; // wave-uniform; once the columns are exhausted every position is final
                    if (parked) parked_levels |= 1u << k;
                    const bool placed = (lane < n && pos < ColIndex) || lane == n;
                    if (lane <= n && (parked || placed))
                    {
                        const size_t col = parked ? (size_t)lane : (size_t)((lane < n) ? pos : n);
#pragma unroll
                        for (int r = 0; r < MD; r++)
                            if (r < dim) out[F + r + col * cap] = hh[r];
                    }
                }
                wave_lds_fence();
                STAMP(6)
                F += dim;
            }

            // ---- solve(): block back-substitution on the compact images (lexlse.h:1015-1045) ----
            if (lane <= NC) xs[lane] = 0.0;
            if (lane <= n) phys_s[(lane < n) ? pos : n] = (uint8_t)lane;
            wave_lds_fence();
            {
                int acc = 0;
                for (int k = nObj; k--;)
                {
                    const int rank = uni((int)meta[4 * k + 1]);
                    if (rank == 0) continue;
                    const int Fc      = uni((int)meta[4 * k + 0]);
                    const double *img = IMG + uni((int)meta[4 * k + 2]);
                    const int w       = n + 1 - Fc;
                    const int c0      = Fc + rank; // == first_col_index of the next level with rank > 0
                    // later column swaps also permuted this level's T block: final position -> physical column -> image column.
                    // Lane j resolves the image column of solved position c0+j and holds x at that position; both reach the row
                    // lanes through SGPRs, so the ordered chain below only waits for its own LDS stream
                    const int src    = (lane < acc) ? (int)phys_s[c0 + lane] : 0;
                    const int myoff  = __builtin_amdgcn_ds_bpermute(src << 2, (int)((slots >> (8 * k)) & 0xffull)) - Fc; // slot of ANOTHER lane's column
                    const double xv  = (lane < acc) ? xs[c0 + lane] : 0.0;
                    const double *row = img + (lane < rank ? lane : 0) * w;
                    // everything the triangular part needs is independent of the chain: fetch it up front
                    double col[MD];
#pragma unroll
                    for (int j = 0; j < MD; j++) col[j] = (j < rank && lane < j) ? row[j] : 0.0;
                    const double dg = row[lane < rank ? lane : 0];
                    double s        = row[n - Fc];
                    for (int j = 0; j < acc; j++)
                    {
                        const int oj    = __builtin_amdgcn_readlane(myoff, j);
                        const double xj = rdlane(xv, j);
                        s               = dfma(-row[oj], xj, s);
                    }
#pragma unroll
                    for (int j = MD; j--;)
                    {
                        if (j >= rank) continue;
                        const double rjj = rdlane(dg, j);
                        const double sj  = rdlane(s, j);
                        const double xj  = sj / rjj;
                        if (lane == j) s = xj;
                        if (lane < j) s = dfma(-col[j], xj, s);
                    }
                    if (lane < rank) xs[Fc + lane] = s;
                    wave_lds_fence();
                    acc += rank;
                }
            }
            STAMP(9)

            // ---- results ----
            if (write_factor) // get_lexqr layout: column = FINAL position of the physical column (lexlse.h:225: swaps span all rows)
            {
                int Fr = 0;
                for (int k = 0; k < nObj; k++)
                {
                    const int dim = (int)dims[k];
                    if ((parked_levels >> k) & 1u)
                    {
                        double blk[MD];
#pragma unroll
                        for (int r = 0; r < MD; r++) blk[r] = (lane <= n && r < dim) ? out[Fr + r + (size_t)lane * cap] : 0.0;
                        // (row r of every column is loaded by ONE instruction and the stores below wait for their data: no column is
                        //  overwritten before it has been read)
                        const int slot = (lane < n) ? pos : n;
                        if (lane <= n && slot != lane)
                        {
#pragma unroll
                            for (int r = 0; r < MD; r++)
                                if (r < dim) out[Fr + r + (size_t)slot * cap] = blk[r];
                        }
                    }
                    else if (dim > 0) // full-rank level: the columns that were still free take their T entries from the image
                    {
                        const int rank    = uni((int)meta[4 * k + 1]);
                        const int Fc      = uni((int)meta[4 * k + 0]);
                        const double *img = IMG + uni((int)meta[4 * k + 2]);
                        const int w       = n + 1 - Fc;
                        const int myslot  = (int)((slots >> (8 * k)) & 0xffull);
                        if (lane < n && myslot >= Fc + rank)
                        {
#pragma unroll
                            for (int r = 0; r < MD; r++)
                                if (r < rank) out[Fr + r + (size_t)pos * cap] = img[r * w + (myslot - Fc)];
                        }
                    }
                    Fr += dim;
                }
            }
            if (lane < n) a.x[(size_t)b * n + lane] = xs[pos]; // x = P x: variable j sits at position pos[j]
            if (lane < n) a.perm[(size_t)b * n + lane] = (lane < TotalRank) ? (uint32_t)perm_s[lane] : (uint32_t)lane;
            if (lane < nObj)
            {
                a.fcol[(size_t)b * nObj + lane] = meta[4 * lane + 0];
                a.rank[(size_t)b * nObj + lane] = meta[4 * lane + 1];
            }
            if (lane == 0) a.totalrank[b] = (uint32_t)TotalRank;
            STAMP(10)
            STAMP_WRITE
        }
    } // namespace

    namespace
    {
        /// exact worst case of sum_k (n+1-Fc_k) * rank_k over rank distributions with rank_k <= md: ranks as large and as early
        /// as possible (the width n+1-Fc_k only shrinks)
        inline uint32_t lwave_image_doubles(uint32_t n, uint32_t nObj, uint32_t md)
        {
            uint32_t fc = 0, total = 0;
            for (uint32_t k = 0; k < nObj && fc < n; k++)
            {
                const uint32_t r = md < n - fc ? md : n - fc;
                total += (n + 1 - fc) * r;
                fc += r;
            }
            return (total + 1) & ~1u;
        }

        template <int NC, int MD, bool EXACT, bool WF>
        hipError_t launch_lwave_t(const LseArgs &a, hipStream_t s)
        {
            const uint32_t img = lwave_image_doubles(a.nVar, a.nObj, MD);
            const size_t lds1  = (8 * ((size_t)img + 2 * (NC + 1) + 6 * MD) + 4 * 4 * (size_t)a.nObj + 2 * 64 + 3 * 64 + 15) & ~(size_t)15; // one wave
            const size_t lds   = lds1 * LEXLS_LWAVE_WPB;
            if (lds > kMaxLdsBytes) return hipErrorInvalidValue;
            if (lds > 64 * 1024)
            {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(lqr_lwave_kernel<NC, MD, EXACT, WF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return e;
            }
            const uint32_t blocks = (a.batch + LEXLS_LWAVE_WPB - 1) / LEXLS_LWAVE_WPB;
            hipLaunchKernelGGL((lqr_lwave_kernel<NC, MD, EXACT, WF>), dim3(blocks), dim3(64 * LEXLS_LWAVE_WPB), lds, s, a, img, (uint32_t)(lds1 / 8));
            return hipGetLastError();
        }
    } // namespace
} // namespace lexls

// One translation unit per instantiation (parallel builds): LEXLS_LWAVE_INSTANCE(name, NC, MD, EXACT, WF)
#define LEXLS_LWAVE_INSTANCE(NAME, NC, MD, EXACT, WF) \
    namespace lexls { hipError_t NAME(const LseArgs &a, hipStream_t s) { return launch_lwave_t<NC, MD, EXACT, WF>(a, s); } }
