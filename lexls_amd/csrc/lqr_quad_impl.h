// FOUR PROBLEMS PER WAVEFRONT, left-looking lexicographic-QR kernel for IK-sized problems (lexlse.h:117-506 + solve() :1015-1045).
//
// Same results, bit for bit, as the other kernels (arithmetic contract of oracle/lexlse_oracle.h): every value sees the same
// ordered fma chain; what changes is the mapping onto the wavefront.  One problem owns one 16-lane DPP row, so
//   * a value that is uniform per problem travels inside the row with ONE v_mov_b64_dpp row_newbcast:<lane> — no SGPR round trip,
//     no LDS broadcast: Householder essentials, tau, Gauss multipliers and the back-substituted x are produced in "lane r <-> row r"
//     form and consumed straight from there;
//   * the pivot search (first maximum by position) is a butterfly inside the row, the per-pivot bookkeeping of lqr_lwave (position
//     map, rank test, scalars: one sqrt + one division sequence) is paid once per FOUR problems;
//   * 4096 problems are 1024 wavefronts = one per SIMD, so a wave may use the whole register file (no occupancy target).
//
// Column layout ("position layout").  Slot s of lane l of a row holds the column whose position in the reference's permuted order
// (lexlse.h:222-232) was P = 16 s + l when the level started; the right-hand side sits at P = n.  A level's rows are loaded from
// HBM in that layout when the level is reached (left-looking: a row's Gauss update by an earlier level depends only on that row
// and on [R_q T_q], lexlse.h:431-471), so the pivot columns of all finished levels — whose positions are final — sit in STATIC
// lanes: pivot c' is lane c' % 16 of slot c' / 16.  The elimination of the level's rows is then one pass over c' = 0 .. Fc-1:
//       l_r = a[r][c'] * (1 / R_c'c')          in every lane; the row-broadcast of lane c' % 16 hands the 12 multipliers to the row
//       a[r][P] = fma(-l_r, U[c'][P], a[r][P])  for every column behind c'   (U[c'][P]: this lane's own entry of the LDS image)
// which is exactly the reference's TRSM + trailing update, per row an ascending chain over the pivots.  The Householder loop only
// touches the slots that still hold live columns (S0 = Fc / 16 and up).
//
// LDS per problem: the compact images [R_q T_q | rhs_q] of the finished levels, row-major, columns kept in CURRENT position order
// (re-packed when a level ends: a column's entries move with the swaps), so eliminations read U[c'][P] at a computed offset and the
// back-substitution reads contiguous rows;  + x by position, a 14-double exchange block for the pivot column, byte maps.
#pragma once
#include "lqr_wave_common.h"

namespace lexls
{
    namespace
    {
        __device__ __forceinline__ void quad_lds_fence()
        {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            asm volatile("" ::: "memory");
        }

        constexpr int kQuadMaxObj = 8; // levels: one byte per level in a column's 64-bit image-index word

        /// SIG: position P sits in slot (P + SIG) / 16, lane (P + SIG) % 16.  With SIG = 16 NS - (n + 1) the right-hand side is the last lane of
        /// the last slot, so the live columns of a level fill the UPPER slots completely and the lower ones drop out of the Householder
        /// loop as early as possible (n = 40: 3, 2, 2, 1 live slots instead of 3, 3, 2, 1); SIG = 0 serves every n <= 16 NS - 1
        /// FIX: fixed variables (lexlse.h:132-156) — own instantiations, so that the common path does not carry them.  The fixed columns take
        /// the first positions by the reference's chained swap rule (an edit of the position -> physical-column map before the first level
        /// is loaded), every level block moves their contribution to its right-hand side when it is loaded (ordered chain over the fixed
        /// positions, row-broadcast operands), and from then on a fixed position behaves like a pivot with a zero reciprocal diagonal.
        template <int NS, int MD, bool WF, int SIG, bool FIX = false>
        __global__ __launch_bounds__(64) void lqr_quad_kernel(LseArgs a, uint32_t img_doubles, uint32_t group_bytes)
        {
            static_assert(NS >= 1 && NS <= 4 && MD <= 16 && (MD % 2) == 0, "shape limits of the row layout");
            extern __shared__ double smem[];
            char *const L  = reinterpret_cast<char *>(smem);
            const int lane = threadIdx.x & 63;
            const int g    = lane >> 4; // row = problem inside the wave
            const int gl   = lane & 15;
            const int n    = (int)a.nVar;
            const int cap  = (int)a.cap;
            const int nObj = (int)a.nObj;
            const uint32_t b  = blockIdx.x * 4u + (uint32_t)g;
            const uint32_t bb = b < a.batch ? b : a.batch - 1u; // rows beyond the batch idle on a valid address
            const bool live   = b < a.batch && !(a.skip && a.skip[bb]);
            const size_t pstride = (size_t)cap * (n + 1);
            const double *in     = a.in + bb * pstride;
            const uint32_t *dims = a.dims + (size_t)bb * nObj;
            double *out          = a.fac + bb * pstride; // (WF) get_lexqr layout: column = final position
            if constexpr (WF)
            {
                if (live)
                    for (int i = gl; i < cap; i += 16) a.hh[(size_t)b * cap + i] = 0.0; // initialize(), lexlse.h:1683
            }
            int parked_levels = 0; // (WF) bit k: the free columns of level k were written by physical column and are put in place at the end

            // ---- LDS carve-up of this row's slice (byte offsets; launch_quad_t computes group_bytes) ----
            const int o_img  = g * (int)group_bytes;
            const int o_xs   = o_img + 8 * (int)img_doubles; // 16*NS : x by position
            const int o_ex   = o_xs + 8 * 16 * NS;           // 2 + 16: [fresh, tail | pivot column] hand-off of a pivot step
            const int o_phys = o_ex + 8 * 18;                // 64 B  : physical column at each position
            const int o_perm = o_phys + 64;                  // 64 B  : column_permutations
            const int o_meta = o_perm + 64;                  // kQuadMaxObj x {first column, rank, image offset, image width}
            const int o_emap = o_meta + 16 * kQuadMaxObj;    // 64 x 8 B: byte k of entry j = column index of PHYSICAL column j in the image of level k
            const int o_dims = o_emap + 512;                 // kQuadMaxObj x u32: this problem's level dimensions
            auto D   = [&](int off) -> double & { return *reinterpret_cast<double *>(L + off); };
            auto D2  = [&](int off) -> double2 & { return *reinterpret_cast<double2 *>(L + off); };
            auto B8  = [&](int off) -> uint8_t & { return *reinterpret_cast<uint8_t *>(L + off); };
            auto U32 = [&](int off) -> uint32_t & { return *reinterpret_cast<uint32_t *>(L + off); };

#pragma unroll
            for (int s = 0; s < 4; s++) B8(o_phys + 16 * s + gl) = (uint8_t)(16 * s + gl);
#pragma unroll
            for (int i = 0; i < 4; i++) D(o_emap + 8 * (16 * i + gl)) = 0.0;
            if (gl < kQuadMaxObj) U32(o_dims + 4 * gl) = (live && gl < nObj) ? dims[gl] : 0u; // one global read; a level's dim then costs an LDS read
#pragma unroll
            for (int s = 0; s < NS; s++) D(o_xs + 8 * (16 * s + gl)) = 0.0;
            quad_lds_fence();

            // The four waves of a CU (one per SIMD) would reach every level's loads together and queue behind each other in the CU's
            // one address unit (a column-per-lane load touches 64 lines); started a little apart they take turns instead
#ifndef LEXLS_QUAD_STAGGER
#define LEXLS_QUAD_STAGGER 20
#endif
            if (LEXLS_QUAD_STAGGER > 0)
            {
                const unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (4 << 6) | ((2 - 1) << 11)); // HW_REG_HW_ID, bits [5:4] = SIMD
                for (unsigned i = 0; i < hwid; i++) __builtin_amdgcn_s_sleep(LEXLS_QUAD_STAGGER);
            }

            double blk[NS][MD]; // the level block, position layout
            double idgreg[NS];  // slot s, lane l: 1 / R_cc of pivot position c = 16 s + l
            int rp[NS];         // slot s, lane l: LDS byte address of the image row of pivot position c
            int rq[NS];         // slot s, lane l: v_perm selector that picks the byte of pivot position c's LEVEL out of a column's index word
            unsigned long long em[NS]; // the index word of the column held in slot s
            int pos[NS];        // current position of the column held in slot s
            int pc[NS];         // its physical column
#pragma unroll
            for (int s = 0; s < NS; s++)
            {
                idgreg[s] = 0.0;
                rp[s]     = o_xs; // (a position that is not a pivot yet "reads" zeros of the x block: see the elimination)
                rq[s]     = 0x0c0c0c00;
                em[s]     = 0ull;
                pos[s]    = 0;
                pc[s]     = 0;
            }

            int ColIndex  = 0; // per row (uniform inside a row), like everything below
            int TotalRank = 0;
            int F         = 0;
            int imgoff    = 0; // doubles
            bool exh      = false;
            int nfix      = 0;
            if constexpr (FIX)
            {
                nfix = (live && a.nfixed) ? (int)a.nfixed[bb] : 0;
                for (int k = gl; k < nfix; k += 16)
                {
                    B8(o_perm + k)    = (uint8_t)a.fixed_idx[(size_t)bb * n + k]; // working copy of fixed_var_index; ends as column_permutations
                    D(o_xs + 8 * k)   = a.fixed_val[(size_t)bb * n + k];          // x.head(nVarFixed) = fixed values (lexlse.h:1384), by position
                }
                quad_lds_fence();
                if (gl == 0)
                    for (int kf = 0; kf < nfix; kf++)
                    {
                        const int coeff = (int)B8(o_perm + kf);
                        for (int j = kf + 1; j < nfix; j++) // the first later entry that refers to position kf now refers to coeff (lexlse.h:146-153)
                            if ((int)B8(o_perm + j) == kf)
                            {
                                B8(o_perm + j) = (uint8_t)coeff;
                                break;
                            }
                        const uint8_t t0    = B8(o_phys + kf); // swap the columns at positions kf and coeff (lexlse.h:141-144)
                        B8(o_phys + kf)     = B8(o_phys + coeff);
                        B8(o_phys + coeff)  = t0;
                    }
                quad_lds_fence();
                ColIndex  = nfix;
                TotalRank = nfix;
                exh       = nfix >= n; // lexlse.h:164-175: nothing left to factorise
            }
            const bool all_fixed = FIX && nfix >= n;
            STAMP_DECL
            STAMP(0)

            for (int k = 0; k < nObj; k++)
            {
                const int dim   = (int)U32(o_dims + 4 * k);
                const bool exh0 = exh;                      // columns exhausted before this level: no Householder loop
                const bool work = dim > 0 && (WF || !exh); // x only: once the columns are exhausted nothing below matters; the factor
                                                           // keeps the multipliers and the eliminated right-hand side of those rows too
                const int Fc    = ColIndex;
                int rank        = 0;

                if (__ballot(work) != 0ull)
                {
                    // =====================================================================================
                    // load the level's rows, position layout
                    // =====================================================================================
                    bool aligned = true;
#pragma unroll
                    for (int s = 0; s < NS; s++)
                    {
                        const int P = 16 * s + gl - SIG;
                        pc[s]       = (P >= 0 && P < n) ? (int)B8(o_phys + P) : n;
                        pos[s]      = (P >= 0 && P <= n) ? P : 0x3fffffff;
                        em[s]       = *reinterpret_cast<const unsigned long long *>(L + o_emap + 8 * pc[s]);
                        aligned     = aligned && (!(work && P >= 0 && P <= n) || (dim == MD && (((F + pc[s] * cap) & 1) == 0)));
                    }
                    if (__ballot(!aligned) == 0ull)
                    {
#pragma unroll
                        for (int s = 0; s < NS; s++)
                        {
                            const bool ld     = work && (16 * s + gl - SIG) >= 0 && (16 * s + gl - SIG) <= n;
                            const double2 *s2 = reinterpret_cast<const double2 *>(in + F + (size_t)pc[s] * cap);
#pragma unroll
                            for (int r = 0; r < MD; r++) blk[s][r] = 0.0;
                            if (ld)
                            {
#pragma unroll
                                for (int r = 0; r < MD / 2; r++)
                                {
#ifdef LEXLS_QUAD_NT_LOADS
                                    typedef double nt_d2 __attribute__((ext_vector_type(2)));
                                    const nt_d2 vv    = __builtin_nontemporal_load(reinterpret_cast<const nt_d2 *>(s2 + r));
                                    const double2 v   = make_double2(vv.x, vv.y);
#else
                                    const double2 v   = s2[r];
#endif
                                    blk[s][2 * r]     = v.x;
                                    blk[s][2 * r + 1] = v.y;
                                }
                            }
                        }
                    }
                    else
                    {
#pragma unroll
                        for (int s = 0; s < NS; s++)
                        {
                            const bool ld     = work && (16 * s + gl - SIG) >= 0 && (16 * s + gl - SIG) <= n;
                            const double *src = in + F + (size_t)pc[s] * cap;
#pragma unroll
                            for (int r = 0; r < MD; r++) blk[s][r] = (ld && r < dim) ? src[r] : 0.0;
                        }
                    }

                    STAMP(1)
#ifdef LEXLS_WAVE_STAMPS
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    STAMP(8)
#endif
                    if constexpr (FIX)
                    {
                        // RHS -= sum_k LOD(:, position k) * x_k over the fixed positions, k ascending (lexlse.h:155); a row of the wavefront with
                        // fewer fixed variables meets zeros in its x block
                        const int nfmax = rows_max(work ? nfix : 0);
                        if (nfmax > 0)
                        {
                            double shift[MD];
#pragma unroll
                            for (int r = 0; r < MD; r++) shift[r] = 0.0;
                            for_each_index<0, 16 * NS - SIG - 1>([&](auto kk) __attribute__((always_inline)) {
                                constexpr int K  = decltype(kk)::value;
                                constexpr int sk = (K + SIG) / 16, lk = (K + SIG) % 16;
                                if (K < nfmax) // wave-uniform
                                {
                                    const double xk = D(o_xs + 8 * K);
#pragma unroll
                                    for (int r = 0; r < MD; r++) shift[r] = dfma(gbc<lk>(blk[sk][r]), xk, shift[r]);
                                }
                            });
#pragma unroll
                            for (int s = 0; s < NS; s++)
                            {
                                const bool isr = (16 * s + gl - SIG) == n;
#pragma unroll
                                for (int r = 0; r < MD; r++) blk[s][r] = sel(isr, blk[s][r] - shift[r], blk[s][r]);
                            }
                        }
                    }
                    // =====================================================================================
                    // Gauss elimination of these rows by every finished pivot c' (lexlse.h:431-471, left-looking)
                    // =====================================================================================
                    const bool wf_aligned = __ballot(work && !(dim == MD && ((F | cap) & 1) == 0)) == 0ull; // (WF) 16-byte stores of a multiplier column
                    const int Fcmax = rows_max(work ? Fc : 0);
                    // Rows that take part all have Fc == Fcmax in a batch of equally shaped, full-rank problems: the pivots then run without
                    // any per-row predicate (a row that does not work in this level may compute garbage, nothing of it is kept).  A divergent
                    // region around the block update would make the compiler copy the whole block at every pivot.
                    const bool elim_uniform = rows_min(work ? Fc : 0x3fffffff) == Fcmax;
                    auto eliminate = [&](auto masked_c) __attribute__((always_inline)) {
                        constexpr bool MASKED = decltype(masked_c)::value;
                        // U[c'][P] of this lane's columns is fetched one pivot ahead of its use (the read depends on a row-broadcast address)
                        auto fetch_u = [&](auto cc, double (&u)[NS]) __attribute__((always_inline)) {
                            constexpr int C  = decltype(cc)::value;
                            constexpr int sc = (C + SIG) / 16, lc = (C + SIG) % 16;
                            const int rowp   = gbci<lc>(rp[sc]);
                            const int selq   = gbci<lc>(rq[sc]);
#pragma unroll
                            for (int s = sc; s < NS; s++)
                            {
                                const unsigned e = __builtin_amdgcn_perm((unsigned)(em[s] >> 32), (unsigned)em[s], (unsigned)selq);
                                u[s]             = D(rowp + 8 * (int)e);
                            }
                        };
                        double ucur[NS], unext[NS];
#pragma unroll
                        for (int s = 0; s < NS; s++) ucur[s] = unext[s] = 0.0;
                        if (Fcmax > 0) fetch_u(std::integral_constant<int, 0>{}, ucur);
                        for_each_index<0, 16 * NS - SIG>([&](auto cc) __attribute__((always_inline)) {
                            constexpr int C  = decltype(cc)::value;
                            constexpr int sc = (C + SIG) / 16, lc = (C + SIG) % 16;
                            if (C < Fcmax) // wave-uniform
                            {
                                if constexpr (C + 1 < 16 * NS - SIG)
                                {
                                    if (C + 1 < Fcmax) fetch_u(std::integral_constant<int, C + 1>{}, unext);
                                }
                                auto body = [&]() __attribute__((always_inline)) {
                                    double lr[MD];
                                    for_each_index<0, MD>([&](auto rr) {
                                        constexpr int r = decltype(rr)::value;
                                        lr[r]           = -gbc<lc>(blk[sc][r] * idgreg[sc]);
                                    });
                                    // (WF) the pivot's own column must survive until the multipliers are written (after the loop): columns at or
                                    // before pivot C absorb -l * 0 — their value is kept (an exact zero may change its sign, nothing else)
                                    double uc[NS];
#pragma unroll
                                    for (int s = sc; s < NS; s++) uc[s] = ucur[s];
                                    if constexpr (WF) uc[sc] = sel(gl > lc, ucur[sc], 0.0);
#pragma unroll
                                    for (int s = sc; s < NS; s++)
                                    {
#pragma unroll
                                        for (int r = 0; r < MD; r++) blk[s][r] = dfma(lr[r], uc[s], blk[s][r]);
                                    }
                                };
                                if constexpr (MASKED)
                                {
                                    if (work && C < Fc) body(); // uniform inside a row
                                }
                                else
                                    body();
#pragma unroll
                                for (int s = 0; s < NS; s++) ucur[s] = unext[s];
                            }
                        });
                    };
                    // ONE form for all rows of the wavefront.  A row whose own pivots end before Fcmax runs the remaining pivot steps with a zero
                    // reciprocal diagonal (idgreg of a position that is not a pivot yet) and the zeros of the x block as "U": zero
                    // multipliers, every fma adds a zero product.  (A per-row guarded form next to this one cost 1.5 % of the x-only kernel
                    // through code size and registers; the factor-keeping kernel stores the multipliers of the row's OWN pivots below.)
#ifdef LEXLS_QUAD_MASKED_ELIMINATION
                    if (elim_uniform)
                        eliminate(std::false_type{});
                    else
                        eliminate(std::true_type{});
#else
                    (void)elim_uniform;
                    eliminate(std::false_type{});
#endif
                    if constexpr (WF)
                    {
                        // the multipliers L = A R^-1 of these rows (lexlse.h:441-446): column c' of the factor, final.  A pivot position's lane
                        // still holds the value its column had when it was pivot c' of the loop above; the product with 1 / R_c'c' is the same one
#pragma unroll
                        for (int s = 0; s < NS; s++)
                        {
                            const int P0 = 16 * s + gl - SIG;
                            if (work && P0 >= 0 && P0 < Fc)
                            {
                                double *dst = out + F + (size_t)P0 * cap;
                                const double sc = (FIX && P0 < nfix) ? 1.0 : idgreg[s]; // (a fixed variable's column stays what it is: lexlse.h keeps it in place)
                                if (wf_aligned)
                                {
#pragma unroll
                                    for (int r = 0; r < MD; r += 2) *reinterpret_cast<double2 *>(dst + r) = make_double2(blk[s][r] * sc, blk[s][r + 1] * sc);
                                }
                                else
                                {
#pragma unroll
                                    for (int r = 0; r < MD; r++)
                                        if (r < dim) dst[r] = blk[s][r] * sc;
                                }
                            }
                        }
                    }

                    STAMP(7)
                    // =====================================================================================
                    // Householder QR with column pivoting of the level (lexlse.h:182-268), slots S0 .. NS-1.
                    // Straight-line per pivot: a row that has stopped (rank found, fewer rows, no work) keeps executing on data nobody reads
                    // again (x only: rows at and below its rank) while its bookkeeping is frozen by selects.
                    // =====================================================================================
                    auto factor_level = [&](auto s0c) __attribute__((always_inline)) {
                        constexpr int S0 = decltype(s0c)::value;
                        double nrm[NS];
#pragma unroll
                        for (int s = S0; s < NS; s++)
                        {
                            nrm[s] = 0.0;
#pragma unroll
                            for (int r = 0; r < MD; r++) nrm[s] = dfma(blk[s][r], blk[s][r], nrm[s]);
                        }
                        bool go = work && !exh0;
                        for_each_index<0, MD>([&](auto cnt) __attribute__((always_inline)) {
                            constexpr int counter = decltype(cnt)::value;
                            const bool act        = go && counter < dim;
                            if (__ballot(act) == 0ull) return;

                            // fresh norm and Householder tail norm (lexlse.h:210-211, :241) are only needed for the pivot column.  One live slot:
                            // computed for every column up front — the chains fill the latency of the search; several live slots: computed on
                            // the selected column after the search (a third of the fma's, ~100 cycles more latency)
                            constexpr bool spec_norms = (NS - S0) == 1;
                            double fr0 = 0.0, tl0 = 0.0;
                            if constexpr (spec_norms)
                            {
#pragma unroll
                                for (int r = counter; r < MD; r++)
                                {
                                    fr0 = dfma(blk[S0][r], blk[S0][r], fr0);
                                    if (r > counter) tl0 = dfma(blk[S0][r], blk[S0][r], tl0);
                                }
                                asm volatile("" : "+v"(fr0), "+v"(tl0)); // stay in front of the search
                            }
                            // pivot: first maximum (by position) of the down-dated norms (lexlse.h:205-206)
                            bool cand[NS];
                            double m = -1.0;
#pragma unroll
                            for (int s = S0; s < NS; s++)
                            {
                                cand[s] = pos[s] >= ColIndex && pos[s] < n;
                                m       = vmax(m, sel(cand[s], nrm[s], -1.0));
                            }
                            m = row_max16(m);
                            unsigned ik[NS], w = 0x7fffffffu;
#pragma unroll
                            for (int s = S0; s < NS; s++)
                            {
                                ik[s] = (cand[s] && nrm[s] == m) ? (((unsigned)pos[s] << 8) | (unsigned)(s << 4) | (unsigned)gl) : 0x7fffffffu;
                                w     = ik[s] < w ? ik[s] : w;
                            }
                            w = row_min16(w);
                            const int ppos = (int)(w >> 8);
                            bool isp[NS], ispany = false;
#pragma unroll
                            for (int s = S0; s < NS; s++)
                            {
                                isp[s] = ik[s] == w && w != 0x7fffffffu;
                                ispany = ispany || isp[s];
                            }
                            // the pivot's lane hands [fresh, tail | column] to its row through LDS
                            {
                                constexpr int ce = counter & ~1;
                                double colv[MD];
#pragma unroll
                                for (int r = ce; r < MD; r++) colv[r] = blk[S0][r];
#pragma unroll
                                for (int s = S0 + 1; s < NS; s++)
                                {
#pragma unroll
                                    for (int r = ce; r < MD; r++) colv[r] = sel(isp[s], blk[s][r], colv[r]);
                                }
                                double frv = fr0, tlv = tl0;
                                if constexpr (!spec_norms)
                                {
#pragma unroll
                                    for (int r = counter; r < MD; r++)
                                    {
                                        frv = dfma(colv[r], colv[r], frv);
                                        if (r > counter) tlv = dfma(colv[r], colv[r], tlv);
                                    }
                                }
                                if (ispany)
                                {
                                    D2(o_ex) = make_double2(frv, tlv);
#pragma unroll
                                    for (int r = ce; r < MD; r += 2) D2(o_ex + 16 + 8 * r) = make_double2(colv[r], colv[r + 1]);
                                }
                            }
                            STAMP(2)
                            quad_lds_fence();
                            const double2 ft   = D2(o_ex);
                            const double fresh = ft.x, tailSq = ft.y;
                            const double c0     = D(o_ex + 16 + 8 * counter);
                            const double spread = D(o_ex + 16 + 8 * (gl < MD ? gl : MD - 1));
                            quad_lds_fence();

                            STAMP(3)
                            const bool cont = act && !(fresh < a.tol); // rank test on the squared norm (lexlse.h:214)
                            go              = sel(act, cont, go);
                            if (__ballot(cont) == 0ull) return;

                            // column "swap": update the position map (lexlse.h:222-232)
#pragma unroll
                            for (int s = S0; s < NS; s++)
                            {
                                const bool front = cont && pos[s] == ColIndex;
                                pos[s]           = sel(front, ppos, pos[s]);
                                pos[s]           = sel(cont && isp[s], ColIndex, pos[s]);
                            }
                            if (cont && gl == 0) B8(o_perm + ColIndex) = (uint8_t)ppos;

                            double idgv;
                            if constexpr (counter < MD - 1)
                            {
                                const bool degenerate = tailSq <= DBL_MIN;
                                double beta           = sqrt(dfma(c0, c0, tailSq));
                                if (c0 >= 0.0) beta = -beta;
                                const double diag   = sel(degenerate, c0, beta);
                                const double den    = c0 - beta;
                                const bool ess_lane = gl > counter && gl < MD;
                                double num          = sel(ess_lane, spread, 1.0);
                                double dnm          = sel(ess_lane, den, diag);
                                if (gl == 0)
                                {
                                    num = beta - c0;
                                    dnm = beta;
                                }
                                const double quo = num / dnm; // lane 0: tau | lanes counter+1 .. MD-1: essential part | the others: 1 / R_jj
                                if constexpr (MD < 16)
                                    idgv = gbc<15>(quo);
                                else
                                    idgv = 1.0 / diag;
                                // (WF: a row of the wavefront that has stopped runs this pivot as an identity reflector — see below)
                                const bool ident  = degenerate || (WF && !cont);
                                const double tau  = sel(ident, 0.0, gbc<0>(quo));
                                const double qe   = sel(ident || !ess_lane, 0.0, quo); // lane r: essential entry of row r
                                const double ntau = -tau;
                                const double et   = qe * ntau;
                                STAMP(4)
                                // x only: H is applied to every column of the live slots — the non-trailing ones hold nothing that is read
                                // again (multipliers / essentials of finished pivots)
                                double e[MD], ett[MD];
                                for_each_index<counter + 1, MD>([&](auto rr) {
                                    constexpr int r = decltype(rr)::value;
                                    e[r]            = gbc<r>(qe);
                                    ett[r]          = gbc<r>(et);
                                });
                                // ONE form for every row of the wavefront.  A row whose reflector is the identity (tau == 0, lexlse.h:239) — or, with the
                                // factor kept, a row that has stopped: its dependent rows are part of the factor — runs the same stream with zero
                                // essentials and a zero tau: every fma adds a zero product, the block comes through unchanged (an exact zero may
                                // change its sign, nothing else can); x only, a stopped row computes on data nobody reads again.  Keeping a second,
                                // select-guarded form next to this one cost 11 % of the x-only kernel through its register pressure alone (825
                                // instead of 127 AGPR moves in the code object; 68.9 -> 61.1 us per 4096 problems).
#pragma unroll
                                for (int s = S0; s < NS; s++)
                                {
                                    double tmp = 0.0;
#pragma unroll
                                    for (int r = counter + 1; r < MD; r++) tmp = dfma(e[r], blk[s][r], tmp);
                                    tmp += blk[s][counter];
                                    blk[s][counter] = dfma(ntau, tmp, blk[s][counter]);
#pragma unroll
                                    for (int r = counter + 1; r < MD; r++) blk[s][r] = dfma(ett[r], tmp, blk[s][r]);
                                }
#pragma unroll
                                for (int s = S0; s < NS; s++) blk[s][counter] = sel(isp[s] && (!WF || cont), diag, blk[s][counter]);
                                if constexpr (WF)
                                {
                                    // essential part and tau leave at once (lane r holds the entry of row r); the column's rows above stay intact
                                    if (cont && gl > counter && gl < dim) out[F + gl + (size_t)ColIndex * cap] = qe;
                                    if (cont && gl == 0) a.hh[(size_t)b * cap + F + counter] = tau;
                                }
                            }
                            else
                            {
                                idgv = 1.0 / c0; // last row of a full level: RemainingRows == 1, no reflector (lexlse.h:239)
                            }
#pragma unroll
                            for (int s = S0; s < NS; s++) idgreg[s] = sel(cont && 16 * s + gl - SIG == ColIndex, idgv, idgreg[s]);

                            ColIndex += cont ? 1 : 0;
                            rank += cont ? 1 : 0;
                            const bool full = cont && ColIndex == n;
                            exh             = exh || full;
                            go              = go && !full;
#pragma unroll
                            for (int s = S0; s < NS; s++) nrm[s] = dfma(-blk[s][counter], blk[s][counter], nrm[s]); // lexlse.h:262-266
                            STAMP(5)
                        });
                    };
                    {
                        const int s0 = (rows_min(work ? Fc : 0x3fffffff) + SIG) >> 4;
                        if (NS > 3 && s0 >= 3)
                            factor_level(std::integral_constant<int, (NS > 3 ? 3 : 0)>{});
                        else if (NS > 2 && s0 >= 2)
                            factor_level(std::integral_constant<int, (NS > 2 ? 2 : 0)>{});
                        else if (NS > 1 && s0 >= 1)
                            factor_level(std::integral_constant<int, (NS > 1 ? 1 : 0)>{});
                        else
                            factor_level(std::integral_constant<int, 0>{});
                    }

                    // =====================================================================================
                    // level end: image [R_k T_k | rhs_k] in end-of-level position order, older images re-packed, maps
                    // =====================================================================================
                    const int wk = n + 1 - Fc;
                    // stores that do not apply go to a dump slot (the hand-off block is idle here): no divergent regions
                    const int dump = o_ex;
                    // (WF) a level of MD rows and full rank in every problem of the wavefront — every level of the IK batch —: its factor rows (the
                    // pivot columns down to their diagonals, the right-hand side) leave from the image below as runs of consecutive lanes, see there
#ifndef LEXLS_QUAD_LEVEL_STORES_BY_COLUMN
                    const bool level_runs = WF && (__ballot(work && !(rank == MD && dim == MD && !exh0 && ((cap | F) & 1) == 0)) == 0ull);
#else
                    const bool level_runs = false;
#endif
#pragma unroll
                    for (int s = 0; s < NS; s++)
                    {
                        const int P0  = 16 * s + gl - SIG;
                        const bool mv = work && P0 <= n && P0 >= Fc; // columns that were live in this level (the RHS included)
                        if (mv)
                        {
                            const int e    = pos[s] - Fc; // index of this column in the level's image: end-of-level position order
                            const int base = o_img + 8 * (imgoff + e);
#pragma unroll
                            for (int p = 0; p < MD; p++) D(sel(p < rank, base + 8 * p * wk, dump)) = blk[s][p];
                            B8(o_emap + 8 * pc[s] + k) = (uint8_t)e;
                            if (P0 < n) B8(o_phys + pos[s]) = (uint8_t)pc[s];
                        }
                        if constexpr (WF)
                        {
                            // factor rows of this level (get_lexqr: column = final position).  Pivot columns of the level: the rows down to
                            // their diagonal are intact and final (essential parts left at pivot time); the right-hand side: position n;
                            // columns still free: rows below the rank (a rank-deficient level only) wait at their CURRENT position — free
                            // positions hold nothing else — and move to the final one at the end (rows above: from the image)
                            const bool parked = work && !exh0 && rank != dim && ColIndex < n;
                            if (parked) parked_levels |= 1 << k;
                            if (!level_runs && work && P0 >= 0 && P0 <= n && (P0 == n || (!exh0 && P0 >= Fc)))
                            {
                                const int pe = pos[s];
                                double *dst  = out + F + (size_t)pe * cap;
                                const int j  = pe - Fc;
                                const bool rhs = P0 == n, pivc = !rhs && j < rank, fr = !rhs && !pivc && parked;
#pragma unroll
                                for (int p = 0; p < MD; p++)
                                    if (p < dim && (rhs || (pivc && p <= j) || (fr && p >= rank))) dst[p] = blk[s][p];
                            }
                        }
                        // pivot position P0 of this level: its image row and the selector of this level's index byte
                        const bool piv = work && P0 >= Fc && P0 < Fc + rank;
                        rp[s]          = sel(piv, o_img + 8 * (imgoff + (P0 - Fc) * wk), rp[s]);
                        rq[s]          = sel(piv, 0x0c0c0c00 | k, rq[s]);
                    }
                    if constexpr (WF)
                    {
                        if (level_runs)
                        {
                            // lane = (column j of the level's MD pivot columns or the right-hand side, pair of rows c): R(2c, j), R(2c+1, j) from the image
                            // just written; a pivot column's rows below its diagonal hold the essential part already (left at pivot time)
                            constexpr int HP = MD / 2;
                            quad_lds_fence();
                            for (int t = 0; t < ((MD + 1) * HP + 15) / 16; t++)
                            {
                                const int idx = 16 * t + gl;
                                const int j   = idx / HP, c = idx - j * HP;
                                if (work && j <= MD)
                                {
                                    const bool rhs = j == MD;
                                    const int e    = rhs ? n - Fc : j;
                                    const int pe   = rhs ? n : Fc + j;
                                    double *dst    = out + F + 2 * c + (size_t)pe * cap;
                                    const double v0 = D(o_img + 8 * (imgoff + (2 * c) * wk + e)), v1 = D(o_img + 8 * (imgoff + (2 * c + 1) * wk + e));
                                    if (rhs || 2 * c + 1 <= j)
                                        *reinterpret_cast<double2 *>(dst) = make_double2(v0, v1);
                                    else if (2 * c == j)
                                        dst[0] = v0;
                                }
                            }
                        }
                    }
                }
                STAMP(6)
                if (gl == 0)
                {
                    U32(o_meta + 16 * k)      = all_fixed ? 0u : (uint32_t)Fc; // (the reference leaves first_col_index at 0 when it returns early)
                    U32(o_meta + 16 * k + 4)  = (uint32_t)rank;
                    U32(o_meta + 16 * k + 8)  = (uint32_t)imgoff;
                    U32(o_meta + 16 * k + 12) = (uint32_t)(n + 1 - Fc);
                }
                quad_lds_fence();
                imgoff += (n + 1 - Fc) * rank;
                TotalRank += rank;
                F += dim;
            }

            // ---- solve(): block back-substitution on the images (lexlse.h:1015-1045); lane p <-> row p of a level ----
#pragma unroll
            for (int s = 0; s < NS; s++)
                if (!(FIX && 16 * s + gl < nfix)) D(o_xs + 8 * (16 * s + gl)) = 0.0; // (by LDS slot, not by position: all 16 NS entries; fixed values stay)
            quad_lds_fence();
            for (int k = nObj; k--;)
            {
                const int rank = live ? (int)U32(o_meta + 16 * k + 4) : 0;
                const int rmax = rows_max(rank);
                if (rmax == 0) continue;
                const int Fc = (int)U32(o_meta + 16 * k), ok = (int)U32(o_meta + 16 * k + 8), wk = (int)U32(o_meta + 16 * k + 12);
                const int c0   = Fc + rank;
                const int acc  = rank > 0 ? TotalRank - c0 : 0;
                const int amax = rows_max(acc);
                const int row  = o_img + 8 * (ok + (gl < rank ? gl : 0) * wk);
                double col[MD];
#pragma unroll
                for (int j = 0; j < MD; j++) col[j] = (j < rank && gl < j) ? D(row + 8 * j) : 0.0;
                const double dg = rank > 0 ? D(row + 8 * (gl < rank ? gl : 0)) : 1.0;
                double sv       = rank > 0 ? D(row + 8 * (n - Fc)) : 0.0;
                // rhs_k - T_k x_later (lexlse.h:1029-1033): the column at final position c sits at its level-k index inside the image;
                // sixteen solved positions per trip — lane j looks up index and x of position c0 + base + j, the row-broadcast hands them out
                for (int base = 0; base < amax; base += 16)
                {
                    const int c     = c0 + base + gl;
                    const bool have = base + gl < acc;
                    const int ph    = have ? (int)B8(o_phys + c) : 0;
                    const int offv  = have ? (int)B8(o_emap + 8 * ph + k) : 0;
                    const double xv = have ? D(o_xs + 8 * c) : 0.0;
                    for_each_index<0, 16>([&](auto jj) {
                        constexpr int j   = decltype(jj)::value;
                        const double uj   = D(row + 8 * gbci<j>(offv));
                        const double tnew = dfma(-uj, gbc<j>(xv), sv);
                        sv                = sel(base + j < acc, tnew, sv);
                    });
                }
                for_each_index<0, MD>([&](auto jj) {
                    constexpr int j = MD - 1 - decltype(jj)::value;
                    if (j < rmax)
                    {
                        const double xj = gbc<j>(sv) / gbc<j>(dg);
                        if (j < rank)
                        {
                            if (gl == j) sv = xj;
                            if (gl < j) sv = dfma(-col[j], xj, sv);
                        }
                    }
                });
                if (gl < rank) D(o_xs + 8 * (Fc + gl)) = sv;
                quad_lds_fence();
            }

            if constexpr (WF)
            {
                // ---- the factor's free columns: T entries from the images, parked rows to their final positions ----
                // (rows are parked by rank-deficient levels only: a wavefront without any neither waits for its factor stores to land before this pass
                // nor between the pass's reads and stores — five drains of the store queue, 45 k of the 270 k cycles of the IK batch)
                const bool any_parked = __ballot(parked_levels != 0) != 0ull;
                if (any_parked) __builtin_amdgcn_s_waitcnt(0); // every factor store of this wavefront has landed before the parked rows are read back
                int Fk = 0;
                for (int k = 0; k < nObj; k++)
                {
                    const int dimk = (int)U32(o_dims + 4 * k);
                    const int Fck = (int)U32(o_meta + 16 * k), rkk = (int)U32(o_meta + 16 * k + 4), okk = (int)U32(o_meta + 16 * k + 8), wkk = (int)U32(o_meta + 16 * k + 12);
                    const bool hhran = live && dimk > 0 && Fck < n;
                    if (__ballot(hhran) != 0ull)
                    {
                        const bool parked = (parked_levels >> k) & 1;
#ifndef LEXLS_QUAD_PLACE_BY_COLUMN
                        // every level of the IK batch: full rank, MD rows.  The T entries of the columns that were free at the level's end go out as runs of
                        // consecutive lanes — lane = (column, pair of rows) in column-major order, 16 lanes = 2 2/3 columns' 96-byte segments — so that a
                        // store instruction writes a few contiguous pieces instead of 64 sixteen-byte ones at a 480-byte stride
                        if ((__ballot(hhran && !(rkk == MD && dimk == MD)) == 0ull) && ((cap | Fk) & 1) == 0 && (MD % 2) == 0)
                        {
                            constexpr int HP = MD / 2;
                            const int f      = hhran ? n - (Fck + rkk) : 0; // free columns when level k ended (uniform in the row)
                            const int trips  = rows_max((f * HP + 15) >> 4);
                            for (int t = 0; t < trips; t++)
                            {
                                const int idx = 16 * t + gl;
                                const int j   = idx / HP, c = idx - j * HP;
                                if (j < f)
                                {
                                    const int P = Fck + rkk + j;
                                    const int e = (int)B8(o_emap + 8 * (int)B8(o_phys + P) + k);
                                    const double v0 = D(o_img + 8 * (okk + (2 * c) * wkk + e)), v1 = D(o_img + 8 * (okk + (2 * c + 1) * wkk + e));
                                    *reinterpret_cast<double2 *>(out + Fk + 2 * c + (size_t)P * cap) = make_double2(v0, v1);
                                }
                            }
                            Fk += dimk;
                            continue;
                        }
#endif
                        double keep[NS][MD];
                        int src[NS];
                        bool mine[NS];
#pragma unroll
                        for (int s = 0; s < NS; s++)
                        {
                            const int P = 16 * s + gl;
                            mine[s]     = hhran && P < n && P >= Fck + rkk; // free when level k ended
                            const int e = mine[s] ? (int)B8(o_emap + 8 * (int)B8(o_phys + P) + k) : 0;
                            src[s]      = Fck + e; // its position when level k ended
#pragma unroll
                            for (int p = 0; p < MD; p++)
                            {
                                double v = 0.0;
                                if (mine[s] && p < rkk)
                                    v = D(o_img + 8 * (okk + p * wkk + e));
                                else if (mine[s] && parked && p < dimk)
                                    v = out[Fk + p + (size_t)src[s] * cap];
                                keep[s][p] = v;
                            }
                        }
                        if (any_parked) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // all parked rows are in registers before any of them is overwritten
                        // (a full-rank level of MD rows — every level of the IK batch — is written as MD / 2 sixteen-byte stores per column instead of
                        // MD masked eight-byte ones: same values, half the store instructions and memory transactions)
                        const bool vec = (__ballot(hhran && !(rkk == MD && dimk == MD)) == 0ull) && ((cap | Fk) & 1) == 0 && (MD % 2) == 0;
#pragma unroll
                        for (int s = 0; s < NS; s++)
                        {
                            const int P = 16 * s + gl;
                            if (mine[s])
                            {
                                if (vec)
                                {
#pragma unroll
                                    for (int p = 0; p < MD; p += 2) *reinterpret_cast<double2 *>(out + Fk + p + (size_t)P * cap) = make_double2(keep[s][p], keep[s][p + 1 < MD ? p + 1 : p]);
                                }
                                else
                                {
#pragma unroll
                                    for (int p = 0; p < MD; p++)
                                        if (p < rkk || (parked && p < dimk)) out[Fk + p + (size_t)P * cap] = keep[s][p];
                                }
                            }
                        }
                    }
                    Fk += dimk;
                }
            }
            STAMP(9)
            // ---- results ----
            if (live)
            {
#pragma unroll
                for (int s = 0; s < NS; s++)
                {
                    const int P = 16 * s + gl; // every position once, whatever the layout offset
                    if (P < n)
                    {
                        a.x[(size_t)b * n + B8(o_phys + P)] = D(o_xs + 8 * P); // x = P x: the variable at position P (lexlse.h:1044)
                        a.perm[(size_t)b * n + P]           = (P < TotalRank) ? (uint32_t)B8(o_perm + P) : (uint32_t)P;
                    }
                }
                if (gl < nObj)
                {
                    a.fcol[(size_t)b * nObj + gl] = U32(o_meta + 16 * gl);
                    a.rank[(size_t)b * nObj + gl] = U32(o_meta + 16 * gl + 4);
                }
                if (gl == 0) a.totalrank[b] = (uint32_t)TotalRank;
            }
            STAMP(10)
            STAMP_WRITE
        }
    } // namespace

    namespace
    {
        /// exact worst case of sum_k (n+1-Fc_k) * rank_k over rank distributions with rank_k <= md
        inline uint32_t quad_image_doubles(uint32_t n, uint32_t nObj, uint32_t md)
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

        template <int NS>
        inline size_t quad_group_bytes(uint32_t n, uint32_t nObj, uint32_t md)
        {
            return (8 * ((size_t)quad_image_doubles(n, nObj, md) + 16 * NS + 18) + 64 + 64 + 16 * kQuadMaxObj + 512 + 4 * kQuadMaxObj + 15) & ~(size_t)15;
        }

        template <int NS, int MD, bool WF, int SIG, bool FIX = false>
        hipError_t launch_quad_t(const LseArgs &a, hipStream_t s)
        {
            const uint32_t img = quad_image_doubles(a.nVar, a.nObj, MD);
            const size_t gbytes = quad_group_bytes<NS>(a.nVar, a.nObj, MD);
            const size_t lds    = 4 * gbytes;
            if (lds > kMaxLdsBytes || a.nObj > (uint32_t)kQuadMaxObj || a.nVar + 1 + SIG > 16u * NS || a.nVar > 63u) return hipErrorInvalidValue;
            if (lds > 64 * 1024)
            {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(lqr_quad_kernel<NS, MD, WF, SIG, FIX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return e;
            }
            const uint32_t blocks = (a.batch + 3u) / 4u;
            hipLaunchKernelGGL((lqr_quad_kernel<NS, MD, WF, SIG, FIX>), dim3(blocks), dim3(64), lds, s, a, img, (uint32_t)gbytes);
            return hipGetLastError();
        }
    } // namespace
} // namespace lexls

// One translation unit per instantiation (parallel builds): LEXLS_QUAD_INSTANCE(name, NS, MD, WF)
#define LEXLS_QUAD_INSTANCE(NAME, NS, MD, WF, SIG) \
    namespace lexls { hipError_t NAME(const LseArgs &a, hipStream_t s) { return launch_quad_t<NS, MD, WF, SIG>(a, s); } }
#define LEXLS_QUAD_INSTANCE_FIX(NAME, NS, MD, WF, SIG) \
    namespace lexls { hipError_t NAME(const LseArgs &a, hipStream_t s) { return launch_quad_t<NS, MD, WF, SIG, true>(a, s); } }
