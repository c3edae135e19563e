// The removal search of a lock-step LexLSI stage as one wavefront per problem (sensitivity_sweep_kernel) — a header because two translation
// units run it: lqr_generic.hip launches it as a kernel of its own, lsi_fused_*.hip runs its body inside the persistent iteration kernel.
#pragma once
#include "lexls_kernels.h"
#include "lexls_launch.h"
#include "lqr_wave_common.h"

namespace lexls
{
    namespace
    {
        /// stage_factor for the first `rows` rows of every column only (a problem of a ragged batch uses fewer rows than the capacity; the
        /// rest of a column is never read by its chains)
        template <int NT>
        __device__ __forceinline__ void stage_factor_rows(const double *__restrict__ G, double *L, uint32_t cap, uint32_t ncol, uint32_t ldl, uint32_t rows, uint32_t tid)
        {
            constexpr uint32_t U = 32; // (the IK shapes in ONE round: 35 rows x 41 columns / 64 lanes = 23 loads per lane)
            const uint32_t total = rows * ncol;
            for (uint32_t base = tid; base < total; base += NT * U)
            {
                double v[U];
#pragma unroll
                for (uint32_t u = 0; u < U; u++)
                {
                    const uint32_t e = base + u * NT;
                    const uint32_t j = e / rows;
                    v[u]             = e < total ? G[(e - j * rows) + (size_t)j * cap] : 0.0;
                }
#pragma unroll
                for (uint32_t u = 0; u < U; u++)
                {
                    const uint32_t e = base + u * NT;
                    if (e < total)
                    {
                        const uint32_t j = e / rows;
                        L[(e - j * rows) + j * ldl] = v[u];
                    }
                }
            }
            __syncthreads();
        }

        // -----------------------------------------------------------------------------------------
        // The removal search of a LexLSI iteration as ONE downward sweep (lexlsi.h:1115-1139 around lexlse.h:611-762).
        //
        // The reference calls ObjectiveSensitivity level by level — objective L costs L + 1 level steps (its own level, then the
        // Householder sequences and the L^T lambda products of the levels above it), so a search that goes through all nObj
        // objectives walks nObj (nObj + 1) / 2 level steps, each a chain of dependent Householder applications.  The multipliers
        // of objective L do not depend on the marks the scan of objective L - 1 left (only findDescentDirection's choice does), so all
        // objectives are swept TOGETHER: level step k serves every objective L >= k at once — objective L = oi + 4 t + rho lives in
        // DPP row rho (register t), lane i of the row is row i of the level, the ordered dot products of a reflector run along the
        // row with v_mov_b64_dpp row_newbcast (no SGPR round trips), the factor is staged in LDS once.  Afterwards the decisions are
        // taken objective by objective in the reference's order (marks carried along, first objective with a wrong-sign multiplier
        // wins) on the stored multipliers, sixteen entries at a time.  Every multiplier is the same ordered chain as in
        // sensitivity_kernel: results are bit-identical (tests/test_gpu_parity.py::test_sensitivity_scan_*).
        // One wavefront per problem; level dims <= 16, at most 8 objectives in a sweep, nVar <= 64.
        // -----------------------------------------------------------------------------------------
        template <int CTRL>
        __device__ __forceinline__ double dpp_min(double v)
        {
            const int lo2 = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
            const int hi2 = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
            const double o = __hiloint2double(hi2, lo2);
            return o < v ? o : v;
        }
        __device__ __forceinline__ double row_minf16(double v)
        {
            v = dpp_min<0xB1>(v);
            v = dpp_min<0x4E>(v);
            v = dpp_min<0x141>(v);
            v = dpp_min<0x140>(v);
            return v;
        }

        constexpr int SWEEP_MD = 16, SWEEP_T = 2; // rows per level; objective registers per DPP row (4 rows x 2 = 8 objectives)

        template <int MD> // rows per level the unrolled reflector loops cover (12 for the IK shapes: a quarter fewer wave-uniform tests than 16)
        __device__ __forceinline__ void sensitivity_sweep_body(const LseArgs &a, const int32_t *obj_index, int32_t obj_all, double tolW, double tolC, int scan_up, const uint32_t b)
        {
            static_assert(MD <= SWEEP_MD, "row layout: one level row per lane of a 16-lane DPP row");
#ifdef LEXLS_SWEEP_STAMPS
            long long sst[6] = {0, 0, 0, 0, 0, 0}, sst0 = clock64();
#define SSTAMP(i) { const long long t_ = clock64(); sst[i] += t_ - sst0; sst0 = t_; }
#else
#define SSTAMP(i)
#endif
            constexpr int TT = SWEEP_T;
            extern __shared__ double smem[];
            const uint32_t lane = threadIdx.x;
            const int rho = (int)(lane >> 4), il = (int)(lane & 15u);
            const uint32_t n = a.nVar, cap = a.cap, nObj = a.nObj;
            const int32_t oi = obj_index ? obj_index[b] : obj_all;
            int32_t *sens    = a.sens + (size_t)b * 3;
            if (oi < 0 || (uint32_t)oi >= nObj)
            {
                if (lane == 0)
                {
                    sens[0]     = 0;
                    sens[1]     = -1;
                    sens[2]     = -2;
                    a.maxabs[b] = 0.0;
                }
                return;
            }
            const int last = scan_up ? (int)nObj - 1 : oi; // objectives oi .. last are swept
            // the level descriptors once into LDS (at most 8 objectives per sweep): the loops below ask for them again and again, and a global
            // load per question is a trip to the vector cache each time (the compiler cannot keep them across the LDS / global stores in between)
            __shared__ uint32_t dims[8], rk[8], fc[8];
            // every small global read of the prologue is issued before the first of them is needed (one trip to memory instead of five in a row)
            const uint32_t g_dim = (lane < 8 && lane < nObj) ? a.dims[(size_t)b * nObj + lane] : 0u;
            const uint32_t g_rk  = (lane < 8 && lane < nObj) ? a.rank[(size_t)b * nObj + lane] : 0u;
            const uint32_t g_fc  = (lane < 8 && lane < nObj) ? a.fcol[(size_t)b * nObj + lane] : 0u;
            const double g_hh0   = lane < cap ? a.hh[(size_t)b * cap + lane] : 0.0;
            const double g_hh1   = lane + 64 < cap ? a.hh[(size_t)b * cap + lane + 64] : 0.0;
            const uint8_t g_ct0  = lane < cap ? a.ctr_type[(size_t)b * cap + lane] : (uint8_t)0;
            const uint8_t g_ct1  = lane + 64 < cap ? a.ctr_type[(size_t)b * cap + lane + 64] : (uint8_t)0;
            const uint8_t g_ft0  = lane < n ? a.fixed_type[(size_t)b * n + lane] : (uint8_t)0;
            if (lane < 8)
            {
                dims[lane] = g_dim;
                rk[lane]   = g_rk;
                fc[lane]   = g_fc;
            }
            __syncthreads();
            const uint32_t nf  = a.nfixed ? a.nfixed[b] : 0;
            const uint32_t ld  = cap | 1u;

            // LDS: staged factor | Householder scalars | multipliers per objective (cap each) | rhs per objective (n each) | fixed-variable
            //      multipliers per objective (n each) | activation types (cap constraint rows, then nVar fixed variables)
            double *Wl     = smem;
            double *hhl    = Wl + (size_t)ld * (n + 1);
            double *LamAll = hhl + cap;
            double *RhsAll = LamAll + (size_t)4 * TT * cap;
            double *FixAll = RhsAll + (size_t)4 * TT * n;
            uint8_t *types = reinterpret_cast<uint8_t *>(FixAll + (size_t)4 * TT * n);
            if (lane < cap) hhl[lane] = g_hh0, types[lane] = g_ct0;
            if (lane + 64 < cap) hhl[lane + 64] = g_hh1, types[lane + 64] = g_ct1;
            for (uint32_t i = 128 + lane; i < cap; i += 64) hhl[i] = a.hh[(size_t)b * cap + i], types[i] = a.ctr_type[(size_t)b * cap + i]; // (capacities beyond 128 rows)
            if (lane < n) types[cap + lane] = g_ft0; // (nVar <= 64 on this kernel)
            for (uint32_t i = lane; i < 4u * TT * (cap + 2 * n); i += 64) LamAll[i] = 0.0;
            {
                uint32_t Mrows = 0; // rows of the levels the sweep can touch
                for (int k = 0; k <= last; k++) Mrows += dims[k];
                stage_factor_rows<64>(a.fac + (size_t)b * cap * (n + 1), Wl, cap, n + 1, ld, Mrows ? Mrows : 1u, lane); // ends with a barrier
            }
            SSTAMP(0)

            // ---- the sweep: level k serves every objective L >= k ----
            uint32_t Fend = 0;
            for (int k = 0; k <= last; k++) Fend += dims[k];
            uint32_t F = Fend;
            for (int k = last; k >= 0; k--)
            {
                const int dim = (int)dims[k], rank = (int)rk[k];
                const uint32_t Fc = fc[k];
                F -= (uint32_t)dim;
                double lam[TT];
                bool act[TT], own[TT];
#pragma unroll
                for (int t = 0; t < TT; t++)
                {
                    const int L = oi + 4 * t + rho;
                    act[t]      = L <= last && L >= k;
                    own[t]      = L == k;
                    double v    = 0.0;
                    if (own[t] && L <= last)
                        v = (il >= rank && il < dim) ? -Wl[F + il + (size_t)n * ld] : 0.0; // residual part of the objective's own level (lexlse.h:658-664)
                    else if (act[t])
                        v = (il < rank) ? RhsAll[(size_t)(L - oi) * n + Fc + il] : 0.0; // lexlse.h:716
                    lam[t] = v;
                }
                // Householder sequence of the level, last reflector first (applyOnTheLeft.m:11-14); all control flow is wave-uniform.
                // The essential parts of ALL the level's reflectors and their scalars are read from LDS up front (one batch of reads instead of
                // one exposed LDS round trip per reflector); tau_J then travels by a row broadcast
                double eall[MD];
#pragma unroll
                for (int J = 0; J < MD; J++) eall[J] = (J < rank && il > J && il < dim) ? Wl[F + il + (size_t)(Fc + J) * ld] : 0.0;
                const double tauv = il < dim ? hhl[F + il] : 0.0;
                for_each_index<0, MD>([&](auto jj) __attribute__((always_inline)) {
                    constexpr int J = MD - 1 - decltype(jj)::value;
                    if (J < rank)
                    {
                        const int rows   = dim - J;
                        const double tau = gbc<J>(tauv);
                        if (rows == 1)
                        {
#pragma unroll
                            for (int t = 0; t < TT; t++) lam[t] = sel(il == J, lam[t] * (1.0 - tau), lam[t]);
                        }
                        else if (tau != 0.0)
                        {
                            const bool tail = il > J && il < dim;
                            const double e  = eall[J];
                            double tt[TT];
#pragma unroll
                            for (int t = 0; t < TT; t++) tt[t] = 0.0;
                            for_each_index<J + 1, MD>([&](auto ii) __attribute__((always_inline)) {
                                constexpr int I = decltype(ii)::value;
                                if (I < dim)
                                {
                                    const double ei = gbc<I>(e);
#pragma unroll
                                    for (int t = 0; t < TT; t++) tt[t] = dfma(ei, gbc<I>(lam[t]), tt[t]);
                                }
                            });
                            const double ce = -(tau * e);
#pragma unroll
                            for (int t = 0; t < TT; t++)
                            {
                                const double tsum = tt[t] + gbc<J>(lam[t]);
                                const double head = dfma(-tau, tsum, lam[t]);
                                const double body = dfma(ce, tsum, lam[t]);
                                lam[t]            = sel(il == J, head, sel(tail, body, lam[t]));
                            }
                        }
                    }
                });
                // keep the level's multipliers (decisions are taken after the sweep)
                SSTAMP(1)
#pragma unroll
                for (int t = 0; t < TT; t++)
                {
                    const int L = oi + 4 * t + rho;
                    if (act[t] && il < dim) LamAll[(size_t)(L - oi) * cap + F + il] = lam[t];
                }
                // rhs.head(Fc) -= L^T lambda (lexlse.h:706-707, :722-728); an objective's own level only if it is not objective 0 (:702)
                if (Fc > 0)
                {
                    double lb[TT][MD];
                    for_each_index<0, MD>([&](auto ii) __attribute__((always_inline)) {
                        constexpr int I = decltype(ii)::value;
#pragma unroll
                        for (int t = 0; t < TT; t++) lb[t][I] = gbc<I>(lam[t]);
                    });
                    for (uint32_t c0 = 0; c0 < Fc; c0 += 16)
                    {
                        const uint32_t c  = c0 + (uint32_t)il;
                        const uint32_t cc = c < Fc ? c : 0;
                        double sacc[TT];
#pragma unroll
                        for (int t = 0; t < TT; t++) sacc[t] = 0.0;
#pragma unroll
                        for (int I = 0; I < MD; I++)
                        {
                            if (I < dim)
                            {
                                const double w = Wl[F + I + (size_t)cc * ld];
#pragma unroll
                                for (int t = 0; t < TT; t++) sacc[t] = dfma(w, lb[t][I], sacc[t]);
                            }
                        }
#pragma unroll
                        for (int t = 0; t < TT; t++)
                        {
                            const int L = oi + 4 * t + rho;
                            if (act[t] && c < Fc && !(own[t] && L == 0)) RhsAll[(size_t)(L - oi) * n + c] -= sacc[t];
                        }
                    }
                }
                __syncthreads();
                SSTAMP(2)
            }

            // ---- fixed variables: lambda_fixed = -LOD[0:nLambda, 0:nf]^T lambda (lexlse.h:742-758) ----
            if (nf > 0)
            {
#pragma unroll
                for (int t = 0; t < TT; t++)
                {
                    const int L = oi + 4 * t + rho;
                    if (L > last) continue;
                    uint32_t nLam = 0;
                    for (int k = 0; k <= L; k++) nLam += dims[k];
                    for (uint32_t c0 = 0; c0 < nf; c0 += 16)
                    {
                        const uint32_t c  = c0 + (uint32_t)il;
                        const uint32_t cc = c < nf ? c : 0;
                        double sacc       = 0.0;
                        const double *lv  = LamAll + (size_t)(L - oi) * cap;
                        for (uint32_t i = 0; i < nLam; i++) sacc = dfma(Wl[i + (size_t)cc * ld], lv[i], sacc);
                        if (c < nf) FixAll[(size_t)(L - oi) * n + c] = -sacc;
                    }
                }
            }
            __syncthreads();

            SSTAMP(3)
            // ---- decisions, objective by objective (findDescentDirection, lexlse.h:935-987): sequential semantics — the most negative
            //      sign-adjusted multiplier wins, the first one among equals; marks are in place before the next group is looked at ----
            double best   = 0.0;
            uint32_t bctr = 0;
            int bobj = -2, found = 0, Lout = oi;
            auto scan_group = [&](uint8_t *ty, const double *lm, uint32_t count, int tag) __attribute__((always_inline)) {
                for (uint32_t g0 = 0; g0 < count; g0 += 16)
                {
                    const uint32_t kx = g0 + (uint32_t)il;
                    const bool in     = kx < count;
                    const uint8_t t   = in ? ty[kx] : (uint8_t)CTR_ACTIVE_EQ;
                    double al         = in ? lm[kx] : 0.0;
                    if (t == CTR_ACTIVE_LB) al = -al;
                    const bool look = in && t != CTR_ACTIVE_EQ && t != CORRECT_SIGN_OF_LAMBDA;
                    if (look && al > tolC && rho == 0) ty[kx] = CORRECT_SIGN_OF_LAMBDA;
                    const bool cand  = look && !(al > tolC) && al < -tolW;
                    const double key = cand ? al : 1.0; // candidates are negative
                    const double m   = row_minf16(key);
                    const unsigned ik = (cand && key == m) ? (unsigned)il : 0xffu;
                    const unsigned fi = row_min16(ik);
                    if (m < 0.0 && m < best) // (wave-uniform: every row holds the same sixteen entries)
                    {
                        best  = m;
                        bctr  = g0 + fi;
                        bobj  = tag;
                        found = 1;
                    }
                }
                // (marks visible to the scans that follow: one wavefront, LDS accesses of a wavefront are served in order — a compiler fence is all)
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                asm volatile("" ::: "memory");
            };
            // The levels of ONE objective are disjoint groups of constraints: their scans do not see each other's marks, only their order matters
            // for ties (own level first, then upwards, strict '<').  So the four DPP rows take FOUR levels at a time — row q the q-th level in scan
            // order — leave {minimum, first index} in LDS, and every lane folds the four in scan order: the same decisions as one level after
            // the other at a quarter of the trips (level dims <= 16 on this kernel: one trip per level).  Fixed variables: as before.
            __shared__ double gm[4];
            __shared__ uint32_t gi[4];
            for (int L = oi; L <= last; L++)
            {
                best  = 0.0;
                bctr  = 0;
                bobj  = -2;
                found = 0;
                Lout  = L;
                const double *lm = LamAll + (size_t)(L - oi) * cap;
                for (int k0 = L; k0 >= 0; k0 -= 4)
                {
                    const int k       = k0 - rho; // this row's level
                    uint32_t Fk       = 0;
                    for (int q = 0; q < k; q++) Fk += dims[q];
                    const uint32_t count = k >= 0 ? dims[k] : 0u;
                    const uint32_t kx    = (uint32_t)il;
                    const bool in        = kx < count;
                    uint8_t *ty          = types + Fk;
                    const uint8_t t      = in ? ty[kx] : (uint8_t)CTR_ACTIVE_EQ;
                    double al            = in ? lm[Fk + kx] : 0.0;
                    if (t == CTR_ACTIVE_LB) al = -al;
                    const bool look = in && t != CTR_ACTIVE_EQ && t != CORRECT_SIGN_OF_LAMBDA;
                    if (look && al > tolC) ty[kx] = CORRECT_SIGN_OF_LAMBDA;
                    const bool cand   = look && !(al > tolC) && al < -tolW;
                    const double key  = cand ? al : 1.0; // candidates are negative
                    const double m    = row_minf16(key);
                    const unsigned ik = (cand && key == m) ? (unsigned)il : 0xffu;
                    const unsigned fi = row_min16(ik);
                    if (il == 0)
                    {
                        gm[rho] = m;
                        gi[rho] = fi;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int q = 0; q < 4; q++)
                    {
                        const double mq = gm[q];
                        if (k0 - q >= 0 && mq < 0.0 && mq < best) // (wave-uniform)
                        {
                            best  = mq;
                            bctr  = gi[q];
                            bobj  = k0 - q;
                            found = 1;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    asm volatile("" ::: "memory");
                }
                if (nf > 0) scan_group(types + cap, FixAll + (size_t)(L - oi) * n, nf, -1);
                if (found) break;
            }

            SSTAMP(4)
            // ---- results of the objective the search stopped at (getWorkspace: [lambda_fixed; lambda], lexlse.h:636-639) ----
            {
                uint32_t nLam = 0;
                for (int k = 0; k <= Lout; k++) nLam += dims[k];
                double *out      = a.lambda + (size_t)b * (n + cap);
                const double *lm = LamAll + (size_t)(Lout - oi) * cap;
                const double *fx = FixAll + (size_t)(Lout - oi) * n;
                for (uint32_t i = lane; i < n + cap; i += 64)
                {
                    double val = 0.0;
                    if (i < nf)
                        val = fx[i];
                    else if (i < nf + nLam)
                        val = lm[i - nf];
                    out[i] = val;
                }
                if (lane == 0)
                {
                    sens[0]     = found;
                    sens[1]     = found ? (int32_t)bctr : -1;
                    sens[2]     = found ? bobj : -2;
                    a.maxabs[b] = best;
                }
            }
            for (uint32_t i = lane; i < cap; i += 64) a.ctr_type[(size_t)b * cap + i] = types[i];
            for (uint32_t i = lane; i < n; i += 64) a.fixed_type[(size_t)b * n + i] = types[cap + i];
            SSTAMP(5)
#ifdef LEXLS_SWEEP_STAMPS
            __syncthreads();
            if (lane == 0)
                for (int i_ = 0; i_ < 6; i_++) a.lambda[(size_t)b * (n + cap) + (n + cap - 6) + i_] = (double)sst[i_];
#endif
        }

        template <int MD>
        __global__ __launch_bounds__(64) void sensitivity_sweep_kernel(LseArgs a, const int32_t *obj_index, int32_t obj_all, double tolW, double tolC, int scan_up)
        {
            sensitivity_sweep_body<MD>(a, obj_index, obj_all, tolW, tolC, scan_up, blockIdx.x);
        }

        /// dynamic LDS of one sweep: staged factor, Householder scalars, multipliers / right-hand sides / fixed-variable multipliers of 8 objectives, types
        inline size_t sweep_lds_bytes(const LseArgs &a)
        {
            return 8 * ((size_t)(a.cap | 1u) * (a.nVar + 1) + a.cap + 8 * ((size_t)a.cap + 2 * a.nVar)) + (((size_t)a.cap + a.nVar + 15) & ~(size_t)15);
        }
    } // namespace
} // namespace lexls
