// Shape-specialised lexicographic-QR kernel for IK-sized problems: ONE WAVEFRONT PER PROBLEM,
// the whole problem in VGPRs.
//
// Applies when n+1 <= 64 columns, nCtr <= 64 rows and every level has <= MD rows (BASELINE
// configs[2]/[3]: n = 40, 5 levels x 12 rows).  It computes exactly what lqr_generic.hip computes
// (same arithmetic contract, oracle/lexlse_oracle.h, hence bit-identical results) but is organised
// around the CDNA4 register file instead of LDS:
//
//   T[j]   (NC doubles / lane)  ROW-PER-LANE image of [A|b]: lane i holds row i, register j holds
//          physical column j.  Rows of lower-priority levels stay here for the whole kernel; the
//          Gauss step updates them with lane-local fma chains (the multipliers L[i,:] never leave
//          the lane that computes them, the level's U rows are LDS broadcast reads).
//   hh[r]  (MD doubles / lane)  COLUMN-PER-LANE image of the level being factorised: lane j holds
//          column j of the level's <= MD rows, so column norms and the Householder dot products
//          are lane-local ordered chains — no cross-lane floating-point reduction anywhere.
//   pos    position of physical column j in the reference's permuted ordering.  Columns are never
//          moved: the reference's column swaps (lexlse.h:222-232) become an update of this map.
//
// Cross-lane traffic per pivot: one DPP max-reduction (row_shr-free butterfly + row_bcast), a few
// v_readlane for the pivot column's scalars, and one LDS exchange that spreads the pivot column
// over the lanes so that the R-1 divisions of makeHouseholderInPlace (plus tau and 1/R_jj) cost ONE
// division sequence.  LDS per wave: the transposition scratch and the compact [R_k T_k | rhs_k]
// images the back-substitution needs (~10 KB), so ~12 waves fit a CU.
#pragma once
#include <cstdlib>
#include "lqr_wave_common.h"
#include "lexls_regularize.h"

// Ragged levels (dim < MD) run the static stream of a full level on a zero-padded block (see factor_level).  The earlier form — the same code
// under run-time guards per element — made `v_mov_b64` the most frequent instruction of the ragged path (the compiler copies a register array
// around every conditional update of one of its elements) and doubled the code (112 KB against a 64 KB instruction cache):
// 1024 LSI-like problems 88.8 -> 78.1 us, lock-step LSI stages 76 -> 64 us.  -DLEXLS_WAVE_UNPADDED brings the guarded form back (A/B builds).
#ifndef LEXLS_WAVE_UNPADDED
#define LEXLS_WAVE_PADDED
#endif
// The pivot's lane hands its column to the wave through LDS and the scalars of the reflector travel by DPP row broadcasts (as in
// lqr_quad_impl.h) instead of ~60 v_readlane / 20 v_writelane per pivot: 78.0 -> 75.5 us per 1024 LSI-like problems.
// -DLEXLS_WAVE_LANE_HANDOFF brings the v_readlane form back (A/B builds).
#ifndef LEXLS_WAVE_LANE_HANDOFF
#define LEXLS_WAVE_LDS_HANDOFF
#endif

namespace lexls
{
    namespace
    {
        // NC: register columns (n+1 <= NC <= 64).  MD: max rows of one level (even).  EXACT: n+1 == NC (n is then a
        // compile-time constant and every column guard folds away).
        //
        // Control flow is kept coarse on purpose: a level whose dim == MD (and, in the Gauss step, whose rank == MD) runs a
        // fully static, branch-free instruction stream (FULL = true); ragged levels take the same code with run-time
        // guards.  Per-element scalar branches around single fma's cost more than the arithmetic they skip.
#ifndef LEXLS_WAVE_OCC
#define LEXLS_WAVE_OCC 2
#endif
        // REG: the regularization family (lexlse.h:277-411) on this kernel: after a level is factorised its transformed right-hand side is
        // damped by lexls_regularize.h's routines — the same ones, in the same order, the generic kernel calls — working on the level's
        // compact LDS image [R T | rhs] as their matrix view; the accumulated null-space basis lives in the handle's scratch and follows
        // the column swaps.  Own instantiations (lqr_small_*_fR.hip): the common path does not carry the calls.
        // The kernel's body is a device function of the problem index: lqr_wave_kernel below runs it once per workgroup, the persistent LexLSI
        // kernel (lsi_fused_impl.h) once per active-set iteration of its instance.
        template <int NC, int MD, bool EXACT, bool WF, bool REG = false>
        __device__ __forceinline__ void lqr_wave_body(const LseArgs &a, uint32_t img_doubles, uint32_t reg_cfg, const uint32_t b)
        {
            constexpr bool write_factor = WF; // factor kept in HBM (get_lexqr / dual solve) or x-only traffic
            extern __shared__ double smem[];
            const int lane       = threadIdx.x;
            const int n          = EXACT ? NC - 1 : (int)a.nVar;
            const int cap        = (int)a.cap;
            const int nObj       = (int)a.nObj;
            const size_t pstride = (size_t)cap * (n + 1);
            if (a.skip && a.skip[b]) return; // uniform per wave

            // ---- LDS carve-up ----
            double *X        = smem;                  // NC*MD : level transposition scratch
            double *EX       = X + NC * MD;           // 128   : [0,64) pivot-column exchange, [64,64+MD) 1/R_qq, [96,104) pivot lanes, [104,120) zeros
            double *idg_s    = EX + 64;               // MD: reciprocal diagonal of the level being factorised
            int *pivl_s      = reinterpret_cast<int *>(EX + 96); // MD: lane of the q-th pivot column
            double *ZB       = EX + 104;              // 16 zeros: what non-trailing columns "read" in the Gauss update
            double *IMG      = EX + 128;              // img_doubles : compact [R T | rhs] images of the levels
            double *xs       = IMG + img_doubles;     // 64    : solution by position
            uint32_t *perm_s = reinterpret_cast<uint32_t *>(xs + 64); // 64
            uint32_t *meta   = perm_s + 64;           // 4*nObj: fc, rank, base, stride
            uint8_t *phys_s  = reinterpret_cast<uint8_t *>(meta + 4 * nObj); // 64: physical column at each final position
            uint8_t *slotmap = phys_s + 64;           // nObj*64: position of each physical column when level k was stored
            uint16_t *offs   = reinterpret_cast<uint16_t *>(slotmap + 64 * nObj); // 64: image offsets of the solved columns
            // REG: the regularization routines' vectors, their work matrix and the null-space basis in LDS behind everything else, as far as
            // the host found room (reg_cfg: bits 0-7 order of the work-matrix window, 0 = none; bit 8 basis in LDS).  What does not fit stays in
            // the handle's scratch, where the generic kernel keeps all of it: same routines, same arithmetic, other address space.
            RegView rv{};
            double *NSp     = nullptr; // the null-space basis as the pivot loop sees it
            uint32_t ldns   = 0;
            double reg_fk   = 0.0;     // lane k: level k's factor
            bool reg_basis  = false;   // types whose damping or least-norm solution reads the basis (lexlse.h:2592: the others never accumulate it)
            if constexpr (REG)
            {
                double *rbase = reinterpret_cast<double *>((reinterpret_cast<uintptr_t>(offs + 64) + 15) & ~(uintptr_t)15);
                rv            = reg_view(a, b, nullptr, 1, 0);
                rv.d          = rbase;
                rv.out        = rv.d + n;
                rv.scal       = rv.out + n;
                rbase         = rv.scal + 8;
                if (a.reg_type == 2 || a.reg_type == 6)
                {
                    rv.cg = rbase;
                    rbase += 10 * n;
                }
                const uint32_t ldl = reg_cfg & 0xffu;
                if (ldl)
                {
                    rv.Dl  = rbase;
                    rv.ldl = ldl;
                    rbase += ldl * ldl;
                }
                if (reg_cfg & 0x100u)
                {
                    rv.NS   = rbase;
                    rv.ldns = (uint32_t)n | 1u;
                }
                NSp       = rv.NS;
                ldns      = rv.ldns;
                reg_fk    = lane < nObj ? a.reg_factor[(size_t)b * nObj + lane] : 0.0; // (fetched here: not a memory latency per level)
                reg_basis = a.reg_type == 1 || a.reg_type == 2 || a.reg_type == 3 || a.reg_type == 8;
            }
            STAMP_DECL

            const uint32_t *dims = a.dims + (size_t)b * nObj;
            int M                = 0;
            for (int k = 0; k < nObj; k++) M += (int)dims[k];

            // ---- prefix reuse (LseArgs::resume_level): levels 0 .. Kres-1 are those of this problem's previous factorization ----
            double *fac_out = a.fac + b * pstride;
            int Kres = 0, Fres = 0; // levels read back, their rows
            uint8_t *rstate = (WF && !REG && a.resume_state) ? a.resume_state + (size_t)b * resume_state_bytes((uint32_t)nObj) : nullptr;
            if (rstate && a.resume_level)
            {
                Kres = uni(a.resume_level[b]);
                Kres = Kres < 0 ? 0 : (Kres > nObj ? nObj : Kres);
                for (int k = 0; k < Kres; k++) Fres += (int)dims[k];
            }

            // ---- load: row-per-lane, coalesced down each column ----
            const double *in = a.in + b * pstride;
            double T[NC];
            if (a.g_cdata) // rows named by reference (a lock-step LSI stage whose problem was formed on the device): no assembled copy
            {
                const uint32_t rl  = a.g_row_ld[(size_t)b * cap + (lane < cap ? lane : 0)];
                const size_t ld    = rl & 0x7fffffffu;
                const double *src  = a.g_cdata + (size_t)b * a.g_per + a.g_row_src[(size_t)b * cap + (lane < cap ? lane : 0)];
                const bool on      = lane < M && ld != 0 && lane >= Fres;
#pragma unroll
                for (int j = 0; j < NC; j++) T[j] = (j <= n && on) ? src[(size_t)(j < n ? j : n + (int)(rl >> 31)) * ld] : 0.0;
            }
            else
            {
#pragma unroll
                for (int j = 0; j < NC; j++) T[j] = (j <= n && lane < M && lane >= Fres) ? in[lane + (size_t)j * cap] : 0.0;
            }
            if (Kres > 0) // the finished rows of the levels read back: from the factor, whose column = the FINAL position of the physical column then
            {
                const int oldpos = (int)rstate[64 * nObj + lane];
#pragma unroll
                for (int j = 0; j < NC; j++)
                    if (j <= n)
                    {
                        const int slot = (j < n) ? __builtin_amdgcn_readlane(oldpos, j) : n;
                        if (lane < Fres) T[j] = fac_out[lane + (size_t)slot * cap];
                    }
            }

            double *hhs = a.hh + (size_t)b * cap;
            for (int i = lane; i < cap; i += 64)
                if (i >= Fres) hhs[i] = 0.0; // initialize(), lexlse.h:1683 (the scalars of levels read back stay)
            perm_s[lane] = lane;
            if (lane < 16) ZB[lane] = 0.0;
            for (uint32_t i = lane; i < img_doubles; i += 64) IMG[i] = 0.0; // (the trailing update of a ragged level reads past a column's rank)

            if constexpr (REG) // initialize(): null_space.setZero() (lexlse.h:1686)
            {
                if (reg_basis)
                    for (int e = lane; e < (int)ldns * (n + 1); e += 64) NSp[e] = 0.0;
            }
            int pos        = (lane < n) ? lane : (lane == n ? n : 0x3fffffff);
            int rowlim     = 64; // factor output: rows of this lane's physical column that the row-per-lane image T still owns (all, until it is pivoted)
            int ColIndex   = 0;
            int TotalRank  = 0;

            // ---- fixed variables (lexlse.h:132-156): their columns take the first positions — an update of the position map
            //      that follows the reference's chained index rule — and their contribution moves to the RHS ----
            const int nf = a.nfixed ? (int)a.nfixed[b] : 0;
            if (nf > 0)
            {
                __syncthreads();
                if (lane < nf)
                {
                    perm_s[lane] = a.fixed_idx[(size_t)b * n + lane]; // working copy of fixed_var_index; ends as column_permutations
                    EX[lane]     = a.fixed_val[(size_t)b * n + lane];
                }
                __syncthreads();
                for (int kf = 0; kf < nf; kf++)
                {
                    const int coeff = uni((int)perm_s[kf]);
                    // the first later entry that refers to position kf now refers to coeff (lexlse.h:146-153)
                    const unsigned long long hit = __ballot(lane > kf && lane < nf && (int)perm_s[lane] == kf);
                    if (hit && lane == (int)__builtin_ctzll(hit)) perm_s[lane] = (uint32_t)coeff;
                    // swap the columns at positions kf and coeff (lexlse.h:141-144)
                    const int la = (int)__builtin_ctzll(__ballot(lane < n && pos == kf));
                    const int lb = (int)__builtin_ctzll(__ballot(lane < n && pos == coeff));
                    if (lane == la) pos = coeff;
                    if (lane == lb) pos = kf;
                    __syncthreads();
                }
                double shift = 0.0; // RHS -= sum_k LOD(:, position k) * x_k, k ascending (lexlse.h:155)
                for (int kf = 0; kf < nf; kf++)
                {
                    const int lk = (int)__builtin_ctzll(__ballot(lane < n && pos == kf));
                    shift        = dfma(select_reg<NC>(T, lk), EX[kf], shift);
                }
                if (lane < Fres) shift = 0.0; // (rows read back from the factor carry it already; x - 0.0 == x)
                if (EXACT)
                    T[NC - 1] -= shift;
                else
                    store_reg<NC>(T, n, select_reg<NC>(T, n) - shift, true);
                ColIndex  = nf;
                TotalRank = nf;
                __syncthreads();
            }
            if (Kres > 0) // the pivots of the levels read back: their entries of column_permutations are still in the output
            {
                const int colK = (int)a.fcol[(size_t)b * nObj + Kres - 1] + (int)a.rank[(size_t)b * nObj + Kres - 1];
                if (lane >= nf && lane < colK) perm_s[lane] = a.perm[(size_t)b * n + lane];
                __syncthreads();
            }
            const bool all_fixed = ColIndex >= n; // lexlse.h:164-175: nothing left to factorise
            uint32_t imgp  = 0; // bump pointer into IMG
            bool exhausted = all_fixed;
            int F          = 0;
            STAMP(0)

            for (int k = 0; k < nObj; k++)
            {
                const int dim_rt = (int)dims[k];
                const int Fc     = all_fixed ? 0 : ColIndex; // the reference leaves first_col_index at 0 when it returns early
                int rank         = 0;
                const bool entered = !exhausted; // the reference reaches this level's body (lexlse.h:164-175, :475-490: it returns / breaks once the columns are exhausted)

                double hh[MD];
#pragma unroll
                for (int r = 0; r < MD; r++) hh[r] = 0.0;

                // =====================================================================================
                // Householder QR with column pivoting of the level (lexlse.h:182-268)
                // =====================================================================================
                auto factor_level = [&](auto full_c) {
                    constexpr bool FULL = decltype(full_c)::value;
#ifdef LEXLS_WAVE_PADDED
                    // PADDED: a ragged level (dim < MD) runs the static, branch-free stream of a full one on a block whose rows >= dim are
                    // zero — every extra fma adds a zero product, the last real row meets a zero tail (tau = 0, R_jj = the entry itself:
                    // lexlse.h:239's "no reflector" outcome).  Only what is structural keeps the run-time dimension: which lanes are the
                    // level's rows, how many pivot steps run, where a pivoted column's factor rows end.
                    static_assert(FULL, "padded levels run the full form");
                    const int dim  = MD;
                    const int dimS = dim_rt;
#else
                    const int dim  = FULL ? MD : dim_rt;
                    const int dimS = dim;
#endif

                    // transpose the level's rows: row-per-lane T -> column-per-lane hh (through LDS)
#ifdef LEXLS_WAVE_PADDED
                    if (lane < MD) // pivots this level does not reach: lane 0 and a zero reciprocal diagonal (zero multipliers in the Gauss step)
                    {
                        pivl_s[lane] = 0;
                        idg_s[lane]  = 0.0;
                    }
#endif
                    __syncthreads();
                    if (lane >= F && lane < F + dimS)
                    {
#pragma unroll
                        for (int j = 0; j < NC; j++)
                            if (j <= n) X[j * MD + (lane - F)] = T[j];
                    }
                    __syncthreads();
                    if (lane <= n)
                    {
#ifdef LEXLS_WAVE_PADDED
#pragma unroll
                        for (int r = 0; r < MD; r++) hh[r] = (r < dimS) ? X[lane * MD + r] : 0.0;
#else
#pragma unroll
                        for (int r = 0; r < MD; r++)
                            if (r < dim) hh[r] = X[lane * MD + r];
#endif
                    }

                    // initial squared norms of the level's columns (lexlse.h:193-196); rows >= dim are zero: fma(0,0,s) == s
                    double nrm = 0.0;
#pragma unroll
                    for (int r = 0; r < MD; r++) nrm = dfma(hh[r], hh[r], nrm);
                    STAMP(1)

                    bool go = !exhausted; // wave-uniform: false once the level hit its rank / the columns ran out
#pragma unroll
                    for (int counter = 0; counter < MD; counter++)
                    {
                        if (!(go && counter < dimS)) continue;
                        const int R   = dim - counter; // compile-time when FULL
                        const int row = F + counter;

#ifdef LEXLS_WAVE_LDS_HANDOFF
                        // (the sums the winner will hand over — every lane on its own column — are independent of the decision: issued ahead of it,
                        //  they run inside the butterfly's latency)
                        double fr = 0.0, tl = 0.0;
#pragma unroll
                        for (int r = 0; r < MD; r++)
                        {
                            if (r >= counter) fr = dfma(hh[r], hh[r], fr);
                            if (r > counter) tl = dfma(hh[r], hh[r], tl);
                        }
#endif
                        // -- pivot: first maximum (by position) of the down-dated norms (lexlse.h:205-206) --
                        //    Decided on the norms' HIGH WORDS first (signed-integer order = the order of non-negative doubles; one v_max_i32_dpp per
                        //    butterfly stage): a unique lane with the largest high word holds the largest norm.  Two lanes with that high word, or
                        //    no non-negative norm among the candidates, take the comparison of whole doubles and the position rule — the same lane either way.
                        const bool cand = (lane < n) && (pos >= ColIndex);
                        const int khi   = cand ? __double2hiint(nrm) : (int)0x80000000;
                        const int mhi   = wave_maxi(khi);
                        int pl;
                        {
                            const unsigned long long mk = __builtin_amdgcn_uicmp((unsigned)khi, (unsigned)mhi, 32 /* == */);
#ifdef LEXLS_WAVE_F64_DECISION
                            if (false)
#else
                            if (mhi >= 0 && __builtin_popcountll(mk) == 1)
#endif
                                pl = (int)__builtin_ctzll(mk);
                            else
                            {
                                const double key     = cand ? nrm : -INFINITY;
                                const double maxv    = wave_max(key);
                                unsigned long long m = __ballot(cand && key == maxv);
                                pl                   = (int)__builtin_ctzll(m);
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
                            }
                        }
                        pl = uni(pl);
                        STAMP(2)

#ifdef LEXLS_WAVE_LDS_HANDOFF
                        // -- fresh norm of the pivot column and the Householder tail norm (lexlse.h:210-211, :241), every lane on its own column;
                        //    the pivot's lane hands [fresh, tail | its column] to the wave through LDS (one write burst, two reads: in-order
                        //    within the wave, no barrier) — it replaced ~60 v_readlane / 20 v_writelane per pivot with their SGPR hazards --
#ifndef LEXLS_WAVE_SCALARS_BY_LDS
                        // the three scalars of the step straight from the pivot's lane (six v_readlane): the square root below starts on them while
                        // the column — needed one entry per lane, only by the division behind the root — is still on its way through LDS
                        {
                            const int ce = counter & ~1; // (folds: the pivot loop is unrolled)
                            if (lane == pl)
                            {
#pragma unroll
                                for (int r = 0; r < MD; r += 2)
                                    if (r >= ce) *reinterpret_cast<double2 *>(EX + 2 + r) = make_double2(hh[r], hh[r + 1]);
                            }
                        }
                        const double fresh = rdlane(fr, pl);
                        const double c0    = rdlane(hh[counter], pl);
                        const double2 ft   = make_double2(fresh, rdlane(tl, pl));
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        asm volatile("" ::: "memory");
                        const int l16       = lane & 15;
                        const double spread = EX[2 + (l16 < MD ? l16 : MD - 1)];
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        asm volatile("" ::: "memory");
#else
                        {
                            const int ce = counter & ~1; // (folds: the pivot loop is unrolled)
                            if (lane == pl)
                            {
                                *reinterpret_cast<double2 *>(EX) = make_double2(fr, tl);
#pragma unroll
                                for (int r = 0; r < MD; r += 2)
                                    if (r >= ce) *reinterpret_cast<double2 *>(EX + 2 + r) = make_double2(hh[r], hh[r + 1]);
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        asm volatile("" ::: "memory");
                        const int l16       = lane & 15;
                        const double2 ft    = *reinterpret_cast<const double2 *>(EX);
                        const double fresh  = ft.x;
                        const double c0     = EX[2 + counter];
                        const double spread = EX[2 + (l16 < MD ? l16 : MD - 1)];
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        asm volatile("" ::: "memory");
#endif
                        if (lane == pl) nrm = fresh;
                        if (fresh < a.tol) // rank test on the squared norm (lexlse.h:214)
                        {
                            go = false;
                            continue;
                        }
                        STAMP(3)

                        // -- column "swap": update the position map (lexlse.h:222-232) --
                        const int ppos = __builtin_amdgcn_readlane(pos, pl);
                        if (lane == 0) perm_s[ColIndex] = (uint32_t)ppos;
                        if constexpr (REG)
                        {
                            if (ppos != ColIndex && reg_basis) // lexlse.h:229-231: the rows of the null-space basis above this level's first column swap too
                            {
                                for (int i = lane; i < Fc; i += 64)
                                {
                                    const double t0                   = NSp[i + (size_t)ColIndex * ldns];
                                    NSp[i + (size_t)ColIndex * ldns] = NSp[i + (size_t)ppos * ldns];
                                    NSp[i + (size_t)ppos * ldns]     = t0;
                                }
                            }
                        }
                        {
                            const unsigned long long mc = __ballot(lane < n && pos == ColIndex);
                            const int lc                = (int)__builtin_ctzll(mc);
                            if (lane == lc) pos = ppos;
                            if (lane == pl) pos = ColIndex;
                            if (write_factor && lane == pl) rowlim = F + dimS; // below this level the column holds Gauss multipliers, stored directly
                        }

                        if (R > 1)
                        {
                            const double tailSq   = ft.y;
                            const bool degenerate = tailSq <= DBL_MIN;
                            double beta           = sqrt(dfma(c0, c0, tailSq));
                            if (c0 >= 0.0) beta = -beta;
                            const double diag = degenerate ? c0 : beta;
                            const double den  = c0 - beta;
                            // every 16-lane row holds the pivot column (lane r of the row: v_r): ONE division sequence gives tau (lane 0 of a
                            // row), the essential part (lanes counter+1 .. dim-1) and 1 / R_jj (the other lanes) in every row, and the DPP row
                            // broadcasts hand them to all lanes
                            const bool ess_lane = l16 > counter && l16 < dim;
                            double num          = ess_lane ? spread : 1.0;
                            double dnm          = ess_lane ? den : diag;
                            if (l16 == 0)
                            {
                                num = beta - c0;
                                dnm = beta;
                            }
                            const double quo = num / dnm;
                            if constexpr (MD < 16)
                            {
                                if (lane == 63) idg_s[counter] = quo; // (lane 15 of a row has no other role)
                            }
                            else
                            {
                                const double rd = 1.0 / diag;
                                if (lane == 63) idg_s[counter] = rd;
                            }
                            STAMP(4)
                            const double tau  = degenerate ? 0.0 : gbc<0>(quo);
                            const double qe   = (degenerate || !ess_lane) ? 0.0 : quo; // lane r of a row: essential entry of row r
                            const double ntau = -tau;
                            double e[MD], ett[MD];
#pragma unroll
                            for (int r = 0; r < MD; r++) e[r] = ett[r] = 0.0;
                            for_each_index<1, MD>([&](auto rc) {
                                constexpr int r = decltype(rc)::value;
                                if (r > counter)
                                {
                                    e[r]   = gbc<r>(qe);
                                    ett[r] = e[r] * ntau;
                                }
                            });

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
                                hh[counter] = dfma(ntau, tmp, hh[counter]);
#pragma unroll
                                for (int r = 0; r < MD; r++)
                                    if (r > counter) hh[r] = dfma(ett[r], tmp, hh[r]);
                            }
                            if (lane == 0) hhs[row] = tau;
                        }
                        else
                        {
                            const double rd = 1.0 / c0;
                            if (lane == 63) idg_s[counter] = rd;
                        }
#else
                        // -- fresh norm of the pivot column and the Householder tail norm (lexlse.h:210-211, :241) --
                        double fr = 0.0, tl = 0.0;
#pragma unroll
                        for (int r = 0; r < MD; r++)
                        {
                            if (r >= counter) fr = dfma(hh[r], hh[r], fr);
                            if (r > counter) tl = dfma(hh[r], hh[r], tl);
                        }
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
                        if (lane == 0) perm_s[ColIndex] = (uint32_t)ppos;
                        if constexpr (REG)
                        {
                            if (ppos != ColIndex && reg_basis) // lexlse.h:229-231: the rows of the null-space basis above this level's first column swap too
                            {
                                for (int i = lane; i < Fc; i += 64)
                                {
                                    const double t0                   = NSp[i + (size_t)ColIndex * ldns];
                                    NSp[i + (size_t)ColIndex * ldns] = NSp[i + (size_t)ppos * ldns];
                                    NSp[i + (size_t)ppos * ldns]     = t0;
                                }
                            }
                        }
                        {
                            const unsigned long long mc = __ballot(lane < n && pos == ColIndex);
                            const int lc                = (int)__builtin_ctzll(mc);
                            if (lane == lc) pos = ppos;
                            if (lane == pl) pos = ColIndex;
                            if (write_factor && lane == pl) rowlim = F + dimS; // below this level the column holds Gauss multipliers, stored directly
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

                            // spread the pivot column over the lanes: lane r gets v_r for the rows r below the pivot row
                            // (v_readlane -> v_writelane pairs: no LDS round trip, no barrier on the pivot's critical path)
                            double num = 1.0;
                            {
                                int nlo = __double2loint(num), nhi = __double2hiint(num);
                                for_each_index<1, MD>([&](auto rc) {
                                    constexpr int r = decltype(rc)::value;
                                    if (r > counter && r < dim)
                                    {
                                        const int slo = __builtin_amdgcn_readlane(__double2loint(hh[r]), pl);
                                        const int shi = __builtin_amdgcn_readlane(__double2hiint(hh[r]), pl);
                                        int tlo = nlo, thi = nhi; // (inline asm cannot bind a by-reference capture directly)
                                        asm("v_writelane_b32 %0, %1, %2" : "+v"(tlo) : "s"(slo), "n"(r));
                                        asm("v_writelane_b32 %0, %1, %2" : "+v"(thi) : "s"(shi), "n"(r));
                                        nlo = tlo;
                                        nhi = thi;
                                    }
                                });
                                num = __hiloint2double(nhi, nlo);
                            }
                            const bool ess_lane = lane > counter && lane < dim;
                            double dnm          = ess_lane ? den : diag; // lanes without a role compute 1/diag (lane 63 is read)
                            if (lane == 0)
                            {
                                num = beta - c0;
                                dnm = beta;
                            }
                            const double quo = num / dnm; // ONE division sequence: tau | essential part | 1/R_jj
                            if (lane == 63) idg_s[counter] = quo;
                            STAMP(4)
                            // wave-uniform tau and essentials (SGPR pairs); zero beyond the level's rows and when H is the identity
                            const double quo_e = (degenerate || !(ess_lane || lane == 0)) ? 0.0 : quo;
                            const double tau   = rdlane(quo_e, 0);
                            double e[MD]; // e[r] = essential entry of row r (absolute row inside the level)
#pragma unroll
                            for (int r = 0; r < MD; r++) e[r] = 0.0;
                            for_each_index<1, MD>([&](auto rc) {
                                constexpr int r = decltype(rc)::value;
                                if (r > counter) e[r] = rdlane(quo_e, r);
                            });

                            // the pivot column now holds beta and the essential part (zeros if degenerate)
                            if (lane == pl)
                            {
                                hh[counter] = diag;
#pragma unroll
                                for (int r = 0; r < MD; r++)
                                    if (r > counter) hh[r] = e[r];
                            }
                            // apply H to the trailing columns and the RHS (lexlse.h:243-246); branch-free: zero essentials are no-ops
                            const bool trailing = ((lane < n) && (pos > ColIndex)) || (lane == n);
                            if (tau != 0.0 && trailing)
                            {
                                double tmp = 0.0;
#pragma unroll
                                for (int r = 0; r < MD; r++)
                                    if (r > counter) tmp = dfma(e[r], hh[r], tmp);
                                tmp += hh[counter];
                                hh[counter] = dfma(-tau, tmp, hh[counter]);
                                double ntau = -tau; // held in a VGPR: "e[r] * ntau" then has a single SGPR operand (one instruction)
                                asm volatile("" : "+v"(ntau));
#pragma unroll
                                for (int r = 0; r < MD; r++)
                                    if (r > counter) hh[r] = dfma(e[r] * ntau, tmp, hh[r]);
                            }
                            if (lane == 0) hhs[row] = tau;
                        }
                        else
                        {
                            if (lane == 63) idg_s[counter] = 1.0 / c0;
                        }
#endif
                        if (lane == 0) pivl_s[counter] = pl;

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
                };

                // A level read back (prefix reuse): its finished rows are in T already (loaded from the factor); what the rest of the loop body
                // needs besides — the column-per-lane block for the compact image, the pivots' lanes and reciprocal diagonals for the Gauss
                // step, the position map as it was after this level — is restored instead of recomputed.  Everything restored is what the
                // factorization left (1 / R_qq is the same correctly rounded quotient), so the rows from level Kres on see the same operands.
                auto replay_level = [&]() {
                    const int rk = uni((int)a.rank[(size_t)b * nObj + k]);
                    pos          = (lane < n) ? (int)rstate[64 * k + lane] : (lane == n ? n : 0x3fffffff);
                    if (lane < MD)
                    {
                        pivl_s[lane] = 0;
                        idg_s[lane]  = 0.0;
                    }
                    __syncthreads();
                    if (lane >= F && lane < F + dim_rt)
                    {
#pragma unroll
                        for (int j = 0; j < NC; j++)
                            if (j <= n) X[j * MD + (lane - F)] = T[j];
                    }
                    __syncthreads();
                    if (lane <= n)
                    {
#pragma unroll
                        for (int r = 0; r < MD; r++) hh[r] = (r < dim_rt) ? X[lane * MD + r] : 0.0;
                    }
                    const int q = pos - Fc;
                    if (lane < n && q >= 0 && q < rk)
                    {
                        double d = 0.0;
#pragma unroll
                        for (int r = 0; r < MD; r++)
                            if (r == q) d = hh[r];
                        pivl_s[q] = lane;
                        idg_s[q]  = 1.0 / d;
                        rowlim    = F + dim_rt;
                    }
                    rank     = rk;
                    ColIndex = Fc + rk;
                    if (ColIndex == n) exhausted = true;
                };
                if (k < Kres)
                {
                    if (dim_rt > 0) replay_level(); // (also a level behind the last column: its rows pass through the block, as in factor_level)
                }
                else if (dim_rt > 0 && (!exhausted || write_factor))
                {
#if defined(LEXLS_WAVE_PADDED)
                    factor_level(std::true_type{});
#elif defined(LEXLS_WAVE_NO_FULL)
                    factor_level(std::false_type{});
#else
                    if (dim_rt == MD)
                        factor_level(std::true_type{});
                    else
                        factor_level(std::false_type{});
#endif
                }
                const int dim = dim_rt;

                const int stride = (rank + 1) & ~1;
                if (lane == 0)
                {
                    meta[4 * k + 0] = (uint32_t)Fc;
                    meta[4 * k + 1] = (uint32_t)rank;
                    meta[4 * k + 2] = imgp;
                    meta[4 * k + 3] = (uint32_t)stride;
                }
                TotalRank += rank;

                // ---- the level's final rows: compact image for the Gauss step / back-substitution ----
                double *img = IMG + imgp;
                if (rank > 0)
                {
                    const int slot = (lane < n) ? pos : n;
                    if (lane <= n && slot >= Fc)
                    {
#pragma unroll
                        for (int r = 0; r < MD; r++)
                            if (r < stride) img[(slot - Fc) * stride + r] = (r < rank) ? hh[r] : 0.0; // padding row zeroed
                    }
                    imgp += (uint32_t)((n + 1 - Fc) * stride);
                }
                if (REG && entered) // lexlse.h:277-411: the level's transformed right-hand side is damped before the Gauss step
                {
                    __syncthreads(); // the image is in place (and so is everything the swaps wrote to the null-space basis)
                    __threadfence_block();
                    // matrix view of the routines: w(F + r, Fc + c) = img[c * stride + r] (columns in position order, the rhs at position n)
                    rv.W  = img - ((size_t)F + (size_t)Fc * (size_t)stride);
                    rv.ld = (size_t)(stride > 0 ? stride : 1);
                    rv.nf     = (uint32_t)nf;
                    rv.fk     = rdlane(reg_fk, k & 63);
                    rv.has_fk = k < 64;
#ifdef LEXLS_WAVE_STAMPS
                    rv.stamp = a.lambda + (size_t)b * (n + cap) + 16;
                    if (lane == 0)
                    {
                        if (k == 0)
                            for (int i_ = 0; i_ < 15; i_++) rv.stamp[i_] = 0.0;
                        rv.stamp[15] = (double)clock64();
                    }
                    __syncthreads();
#endif
                    regularize_level<64>(a, b, rv, (uint32_t)k, (uint32_t)F, (uint32_t)Fc, (uint32_t)rank, (uint32_t)(n - ColIndex), (uint32_t)lane);
                    __syncthreads();
                    if (rank > 0 && lane == n) // the damped right-hand side is what the factor keeps (and what the Gauss step below subtracts)
                    {
#pragma unroll
                        for (int r = 0; r < MD; r++)
                            if (r < rank) hh[r] = img[(n - Fc) * stride + r];
                    }
                }
                slotmap[k * 64 + lane] = (uint8_t)((lane < n) ? pos : n);
                if (write_factor && dim > 0) // the level's final rows go back into the row-per-lane image (factor output)
                {
                    __syncthreads();
                    if (lane <= n)
                    {
#pragma unroll
                        for (int r = 0; r < MD; r++)
                            if (r < dim) X[lane * MD + r] = hh[r];
                    }
                    __syncthreads();
                    if (lane >= F && lane < F + dim)
                    {
#pragma unroll
                        for (int j = 0; j < NC; j++)
                            if (j <= n) T[j] = X[j * MD + (lane - F)];
                    }
                }
                __syncthreads();
                STAMP(6)

                // =====================================================================================
                // Gauss step on the rows below (lexlse.h:431-471): lane i = row i
                // =====================================================================================
                auto gauss = [&](auto full_c) {
                    constexpr bool FULL = decltype(full_c)::value;
                    const int rk        = FULL ? MD : rank;
                    const int Fn        = F + dim;
                    const bool below    = lane >= Fn && lane < M && lane >= Fres; // (rows of levels read back are final)

                    // ---- L <- A_left R^-1, column-oriented: after L_p is final, ONE batch of loads brings row p of R and every
                    //      later column absorbs it (same accumulation order per column as the row-oriented form: p ascending) ----
                    double acc[MD], Lv[MD], idg[MD];
                    int pq[MD];
#pragma unroll
                    for (int q = 0; q < MD; q++)
                    {
                        const bool on = FULL || q < rk;
                        pq[q]         = on ? uni(pivl_s[q]) : 0;
                        idg[q]        = on ? idg_s[q] : 0.0;
                        acc[q]        = on ? select_reg<NC>(T, pq[q]) : 0.0;
                        Lv[q]         = 0.0;
                    }
#pragma unroll
                    for (int p2 = 0; p2 < MD; p2++)
                    {
                        if (FULL || p2 < rk)
                        {
                            Lv[p2] = acc[p2] * idg[p2];
                            double rrow[MD];
#pragma unroll
                            for (int q = 0; q < MD; q++)
                                if (q > p2) rrow[q] = (FULL || q < rk) ? img[q * stride + p2] : 0.0; // R[p2][q], wave-uniform
#pragma unroll
                            for (int q = 0; q < MD; q++)
                                if (q > p2) acc[q] = dfma(-Lv[p2], rrow[q], acc[q]);
                            // factor output: the multipliers go straight to their final place (pivot p2 of this level sits at position Fc + p2;
                            // lanes = consecutive rows: one coalesced store) instead of into T through a dynamic register index
#ifdef LEXLS_WAVE_PADDED
                            if (write_factor && below && p2 < rank) fac_out[lane + (size_t)(Fc + p2) * cap] = Lv[p2];
#else
                            if (write_factor && below) fac_out[lane + (size_t)(Fc + p2) * cap] = Lv[p2];
#endif
                        }
                    }
#pragma unroll
                    for (int q = 0; q < MD; q++) Lv[q] = below ? -Lv[q] : 0.0; // rows that are not below this level must not change
                    STAMP(7)

                    // ---- Trailing += (-L) * Up over the still-free columns and the RHS, in chunks of CH physical columns:
                    //      pivot index outermost inside a chunk, so the CH fma chains advance side by side (no chain waits for
                    //      its own previous link) and the broadcast loads of U stream ahead.  Columns that are not trailing read
                    //      a block of zeros: their row-per-lane entries are left untouched without any branch. ----
                    const unsigned long long tmask = __ballot(((lane < n) && (pos >= ColIndex)) || (lane == n));
#ifndef LEXLS_GEMM_CH
#define LEXLS_GEMM_CH 1
#endif
                    constexpr int CH               = LEXLS_GEMM_CH;
                    // A ragged level (rank < MD) runs the same straight-line stream over HP <= MD / 2 pivot pairs: pairs past the rank read
                    // whatever follows the column in the image (the next column, or the zero-initialised slack behind the last one: finite)
                    // and multiply it by the zero multipliers Lv[q >= rank] — an exact no-op, like the rows that are not below — instead of
                    // a branch per pair, which would put every broadcast read on the critical path.
                    auto trailing = [&](auto hp_c) __attribute__((always_inline)) {
                        constexpr int HP = decltype(hp_c)::value;
#pragma unroll
                        for (int j0 = 0; j0 < NC; j0 += CH)
                        {
                            if (((tmask >> j0) & ((1ull << CH) - 1ull)) == 0ull) continue; // no trailing column in this chunk
                            const double2 *u[CH];
#pragma unroll
                            for (int c = 0; c < CH; c++)
                            {
                                const int j = j0 + c;
                                u[c]        = reinterpret_cast<const double2 *>(ZB);
                                if (j < NC && j <= n)
                                {
                                    const int slot = (j < n) ? __builtin_amdgcn_readlane(pos, j < 64 ? j : 0) : n;
                                    if ((tmask >> j) & 1ull) u[c] = reinterpret_cast<const double2 *>(img + (slot - Fc) * stride);
                                }
                            }
#pragma unroll
                            for (int pp = 0; pp < HP; pp++)
                            {
                                double2 uv[CH];
#pragma unroll
                                for (int c = 0; c < CH; c++) uv[c] = u[c][pp];
#pragma unroll
                                for (int c = 0; c < CH; c++)
                                    if (j0 + c < NC)
                                    {
                                        T[j0 + c < NC ? j0 + c : 0] = dfma(Lv[2 * pp], uv[c].x, T[j0 + c < NC ? j0 + c : 0]);
                                        T[j0 + c < NC ? j0 + c : 0] = dfma(Lv[2 * pp + 1], uv[c].y, T[j0 + c < NC ? j0 + c : 0]);
                                    }
                            }
                        }
                    };
#ifdef LEXLS_WAVE_PADDED
                    if (stride > MD / 2 + (MD / 2) % 2)
#else
                    if (FULL || stride > MD / 2 + (MD / 2) % 2)
#endif
                        trailing(std::integral_constant<int, MD / 2>{});
                    else
                        trailing(std::integral_constant<int, (MD / 2 + 1) / 2>{}); // rank <= MD / 2 (rounded up to a pair): half the stream
                    STAMP(8)
                };
                if (k + 1 < nObj && rank > 0)
                {
#if defined(LEXLS_WAVE_PADDED)
                    gauss(std::true_type{});
#elif defined(LEXLS_WAVE_NO_FULL)
                    gauss(std::false_type{});
#else
                    if (rank == MD)
                        gauss(std::true_type{});
                    else
                        gauss(std::false_type{});
#endif
                }
                F += dim;
            }

            // ---- solve(): block back-substitution on the compact images (lexlse.h:1015-1045) ----
            __syncthreads();
            xs[lane] = (lane < nf) ? a.fixed_val[(size_t)b * n + lane] : 0.0; // x.head(nVarFixed) = fixed values (lexlse.h:1384)
            if (lane <= n) phys_s[(lane < n) ? pos : n] = (uint8_t)lane;
            __syncthreads();
            {
                int acc = 0;
                for (int k = nObj; k--;)
                {
                    const int rank = uni((int)meta[4 * k + 1]);
                    if (rank == 0) continue;
                    const int Fc       = uni((int)meta[4 * k + 0]);
                    const double *img  = IMG + uni((int)meta[4 * k + 2]);
                    const int stride   = uni((int)meta[4 * k + 3]);
                    const int c0       = Fc + rank; // == first_col_index of the next level with rank > 0
                    // later column swaps also permuted this level's T block: final position -> physical column -> slot at store time
                    // (lane j resolves the offset of position c0 + j once; the ordered fma chain below then only streams)
                    if (lane < acc) offs[lane] = (uint16_t)(((int)slotmap[k * 64 + phys_s[c0 + lane]] - Fc) * stride);
                    __syncthreads();
                    double s = 0.0;
                    if (lane < rank)
                    {
                        s = img[(n - Fc) * stride + lane];
#pragma unroll 4
                        for (int j = 0; j < acc; j++) s = dfma(-img[offs[j] + lane], xs[c0 + j], s);
                    }
                    for (int j = rank; j--;)
                    {
                        const double rjj = img[j * stride + j]; // uniform address
                        const double sj  = rdlane(s, j);
                        const double xj  = sj / rjj;
                        if (lane == j) s = xj;
                        if (lane < j) s = dfma(-img[j * stride + lane], xj, s);
                    }
                    if (lane < rank) xs[Fc + lane] = s;
                    __syncthreads();
                    acc += rank;
                }
            }
            STAMP(9)

            // ---- results ----
            if constexpr (REG) // the null-space basis goes where solveLeastNorm_3 reads it (lexlse.h:93): the handle's scratch, ld = n
            {
                if (reg_basis && (reg_cfg & 0x100u))
                {
                    double *NSg = a.reg_scratch + (size_t)b * reg_scratch_doubles((uint32_t)n);
                    for (int e = lane; e < n * (n + 1); e += 64) NSg[e] = NSp[(e % n) + (size_t)(e / n) * ldns];
                }
            }
            if (write_factor) // get_lexqr layout: column = FINAL position of the physical column
            {
#pragma unroll
                for (int j = 0; j < NC; j++)
                    if (j <= n)
                    {
                        const int slot = (j < n) ? __builtin_amdgcn_readlane(pos, j) : n;
                        const int lim  = (j < n) ? __builtin_amdgcn_readlane(rowlim, j) : 64; // (the multipliers below a pivoted column are in place already)
                        if (lane < M && lane < lim) fac_out[lane + (size_t)slot * cap] = T[j];
                    }
            }
            if (rstate) // what a later factorization needs to read levels back (LseArgs::resume_level)
            {
                for (int k = 0; k < nObj; k++) rstate[64 * k + lane] = slotmap[k * 64 + lane];
                rstate[64 * nObj + lane] = (uint8_t)((lane < n) ? pos : n);
            }
            if (lane < n) a.x[(size_t)b * n + lane] = xs[pos]; // x = P x: variable j sits at position pos[j]
            if (lane < n) a.perm[(size_t)b * n + lane] = (lane < TotalRank) ? perm_s[lane] : (uint32_t)lane;
            if (lane < nObj)
            {
                a.fcol[(size_t)b * nObj + lane] = meta[4 * lane + 0];
                a.rank[(size_t)b * nObj + lane] = meta[4 * lane + 1];
            }
            if (lane == 0) a.totalrank[b] = (uint32_t)TotalRank;
            STAMP(10)
            STAMP_WRITE
        }

        template <int NC, int MD, bool EXACT, bool WF, bool REG = false>
        __global__ __launch_bounds__(64, LEXLS_WAVE_OCC) void lqr_wave_kernel(LseArgs a, uint32_t img_doubles, uint32_t reg_cfg)
        {
            lqr_wave_body<NC, MD, EXACT, WF, REG>(a, img_doubles, reg_cfg, blockIdx.x);
        }
    } // namespace

    namespace
    {
        /// doubles of the compact level images: worst case of sum_k (n+1-Fc_k) * even(rank_k) over rank distributions with rank_k <= MD
        template <int MD>
        inline uint32_t wave_img_doubles(const LseArgs &a)
        {
            const uint32_t n = a.nVar;
            return (n * n) / 2 + n + (n * MD) / 2 + a.nObj * (n + 1) + 64 + MD * MD; // (+ zeros behind the last image: a padded level reads MD columns of it)
        }
        template <int NC, int MD>
        inline size_t wave_lds_bytes(const LseArgs &a, uint32_t img)
        {
            return 8 * ((size_t)NC * MD + 128 + img + 64) + 4 * (64 + 4 * (size_t)a.nObj) + 64 + 64 * (size_t)a.nObj + 128 + 16;
        }

        template <int NC, int MD, bool EXACT, bool WF, bool REG = false>
        hipError_t launch_wave_t2(const LseArgs &a, hipStream_t s)
        {
            const uint32_t n   = a.nVar;
            const uint32_t img = wave_img_doubles<MD>(a);
            size_t lds         = wave_lds_bytes<NC, MD>(a, img);
            uint32_t reg_cfg   = 0;
            if constexpr (REG)
            {
                // the routines' vectors always; then their work matrix (order bounded by the type: the damped triangle alone for R / R_NO_Z /
                // RT_NO_Z, under n/2 + the largest level for TIKHONOV — tikhonov_2 is taken while Fc + rank <= n/2, tikhonov_1 has order n - Fc —,
                // n for TIKHONOV_2) and the null-space basis, each while LEXLS_REG_LDS_WAVES wavefronts (default 8: the occupancy is worth more than either, measured) still share a CU's LDS
                static const size_t share = [] {
                    const char *e = std::getenv("LEXLS_REG_LDS_WAVES");
                    const long w  = e ? std::atol(e) : 8;
                    return kMaxLdsBytes / (size_t)(w < 1 ? 1 : (w > 8 ? 8 : w));
                }();
                const bool cg = a.reg_type == 2 || a.reg_type == 6;
                lds           = ((lds + 15) & ~(size_t)15) + 16 + 8 * (2 * (size_t)n + 8 + (cg ? 10 * (size_t)n : 0));
                uint32_t order = 0;
                switch (a.reg_type)
                {
                case 3: case 4: case 5: order = MD; break;
                case 1: order = n / 2 + MD + 1 < n ? n / 2 + MD + 1 : n; break;
                case 8: order = n; break;
                default: break;
                }
                if (order > 255) order = 0;
                if (order && lds + 8 * (size_t)order * order <= share)
                {
                    reg_cfg |= order;
                    lds += 8 * (size_t)order * order;
                }
                const bool basis = a.reg_type == 1 || a.reg_type == 2 || a.reg_type == 3 || a.reg_type == 8;
                if (basis && lds + 8 * (size_t)(n | 1u) * (n + 1) <= share)
                {
                    reg_cfg |= 0x100u;
                    lds += 8 * (size_t)(n | 1u) * (n + 1);
                }
            }
            if (lds > kMaxLdsBytes) return hipErrorInvalidValue;
            if (lds > 64 * 1024)
            {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(lqr_wave_kernel<NC, MD, EXACT, WF, REG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return e;
            }
            if (REG && !a.reg_scratch) return hipErrorInvalidValue;
            hipLaunchKernelGGL((lqr_wave_kernel<NC, MD, EXACT, WF, REG>), dim3(a.batch), dim3(64), lds, s, a, img, reg_cfg);
            return hipGetLastError();
        }

    } // namespace
} // namespace lexls

// One translation unit per instantiation (parallel builds): LEXLS_WAVE_INSTANCE(name, NC, MD, EXACT, WF)
#define LEXLS_WAVE_INSTANCE(NAME, NC, MD, EXACT, WF) \
    namespace lexls { hipError_t NAME(const LseArgs &a, hipStream_t s) { return launch_wave_t2<NC, MD, EXACT, WF>(a, s); } }
#define LEXLS_WAVE_INSTANCE_REG(NAME, NC, MD) \
    namespace lexls { hipError_t NAME(const LseArgs &a, hipStream_t s) { return launch_wave_t2<NC, MD, false, true, true>(a, s); } }
