// Device side of the lock-step LexLSI driver (lexls_lsi_capi.hip): the step of an iteration and the resident working-set iteration, one wavefront per
// instance.  A header because two translation units run it: the driver launches lsi_step_kernel / lsi_iterate_kernel as kernels of their own,
// lsi_fused_*.hip runs the iteration's body inside the persistent kernel (l-QR -> removal sweep -> iteration, until the instance stops).
#pragma once
#include <lexls/lexls.h>
#include <hip/hip_runtime.h>
#include "lqr_wave_common.h" // wave_max

namespace
{
    using namespace LexLS;

    // =============================================================================================
    // The step of one active-set iteration on the device (lexlsi.h:987-1029 + :1234-1240, objective.h:260-338, :521-589)
    // =============================================================================================
    constexpr uint32_t STEP_MAX_OBJ = 16;
    struct StepShape
    {
        uint32_t n, nObj, total, SD; // SD = n + 2 total: per instance [x | v | A x]
        uint32_t dim[STEP_MAX_OBJ], simple[STEP_MAX_OBJ], first[STEP_MAX_OBJ];
        uint64_t off[STEP_MAX_OBJ]; // first element of the objective's [A | lb | ub] (or [lb | ub]) block inside a problem's data
        uint64_t per_data;
        uint32_t dim0;
        double tol_feasibility;
    };
    /// working-set block of a stage: per instance `mode` (0: no step, 1: state on the device, 2: state arrives in the staging copy),
    /// per constraint its activation type (0 = inactive) and, for the inactive ones, the position in the objective's inactive list
    struct StepArgs
    {
        StepShape sh;
        uint32_t B;
        const double *cdata;      // B x per_data
        const uint32_t *var;      // B x dim0: variable indices of a simple-bounds objective 0
        const double *x_lse;      // B x n: solution of the equality problem
        double *state;            // B x SD
        const double *state_in;   // B x SD staging (mode 2)
        const uint8_t *mode;      // B
        const uint8_t *ctr_state; // B x total
        const uint16_t *inact_pos; // B x total
        double *res;              // B x 4: alpha, blocking objective (-1: none), constraint, type
    };

    /// One wavefront, one instance: dx = x_lse - x, A*dx, dv, the ratio test over the inactive constraints (first minimum in working-set
    /// scan order) and the update of x / v / A*x in `st` (read from `src`).  Every lane returns the verdict.
    __device__ __forceinline__ void lsi_step_wave(const StepShape &sh, const double *data, const uint32_t *var_b, const double *x_lse_b, const double *src, double *st,
                                                  const uint8_t *ctr_state_b, const uint16_t *inact_pos_b, double *dx_s, double *adx_s, double *dv_s, double &alpha,
                                                  int &blk_obj, uint32_t &blk_ctr, uint32_t &blk_type)
    {
        const uint32_t lane = threadIdx.x & 63u;
        const uint32_t n = sh.n, total = sh.total;
        // dx = x_lse - x (lexlsi.h:990-991)
        for (uint32_t j = lane; j < n; j += 64) dx_s[j] = x_lse_b[j] - src[j];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        asm volatile("" ::: "memory");

        double best = INFINITY; // smallest ratio below 1 this lane has seen, first in scan order on ties
        uint32_t best_key = 0xffffffffu, best_ctr = 0, best_type = 0;
        for (uint32_t g = lane; g < total; g += 64) // lane = constraint (all objectives side by side)
        {
            uint32_t k = 0;
            while (k + 1 < sh.nObj && g >= sh.first[k + 1]) k++;
            const uint32_t dim = sh.dim[k], c = g - sh.first[k];
            const double *blk  = data + sh.off[k];
            double adx, lb, ub;
            if (sh.simple[k])
            {
                adx = dx_s[var_b[c]]; // objective.h:266-270
                lb  = blk[c];
                ub  = blk[c + dim];
            }
            else
            {
                adx = 0.0; // one ordered chain per row, as the host's apply_A; twenty loads in flight (the chain is short, the loads are not)
                uint32_t j = 0;
                for (; j + 20 <= n; j += 20)
                {
                    double av[20];
#pragma unroll
                    for (int u = 0; u < 20; u++) av[u] = blk[c + (size_t)(j + u) * dim];
#pragma unroll
                    for (int u = 0; u < 20; u++) adx = lexls::dfma(av[u], dx_s[j + u], adx);
                }
                for (; j + 4 <= n; j += 4)
                {
                    double av[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) av[u] = blk[c + (size_t)(j + u) * dim];
#pragma unroll
                    for (int u = 0; u < 4; u++) adx = lexls::dfma(av[u], dx_s[j + u], adx);
                }
                for (; j < n; j++) adx = lexls::dfma(blk[c + (size_t)j * dim], dx_s[j], adx);
                lb = blk[c + (size_t)n * dim];
                ub = blk[c + (size_t)(n + 1) * dim];
            }
            const double v = src[n + g], ax = src[n + total + g];
            const uint32_t act = ctr_state_b[g];
            double dv = -v; // objective.h:292
            if (act)
            {
                const double rhs = (act == CTR_ACTIVE_LB) ? lb : ub;
                dv               = dv + ((ax + adx) - rhs); // :300-330
            }
            else
            {
                const double den = adx - dv; // objective.h:532-569
                uint32_t type    = 0;
                double rhs       = 0.0;
                if (den < -sh.tol_feasibility)
                {
                    type = CTR_ACTIVE_LB;
                    rhs  = lb;
                }
                else if (den > sh.tol_feasibility)
                {
                    type = CTR_ACTIVE_UB;
                    rhs  = ub;
                }
                if (type)
                {
                    const double num = (rhs - ax) + v;
                    double ratio     = num / den;
                    if (ratio < 0.0) ratio = 0.0;
                    const uint32_t key = (k << 16) | inact_pos_b[g];
                    if (ratio < 1.0 && (ratio < best || (ratio == best && key < best_key)))
                    {
                        best      = ratio;
                        best_key  = key;
                        best_ctr  = c;
                        best_type = type;
                    }
                }
            }
            adx_s[g] = adx;
            dv_s[g]  = dv;
        }
        // first minimum in scan order over the wave (strict '<' of the sequential scan, lexlsi.h:1011-1019)
        const double wmin = -lexls::wave_max(-best);
        alpha             = 1.0;
        blk_obj           = -1;
        blk_ctr = 0, blk_type = 0;
        if (wmin < 1.0)
        {
            unsigned long long tied = __ballot(best == wmin);
            uint32_t kmin           = 0xffffffffu;
            int win                 = 0;
            while (tied)
            {
                const int l = (int)__builtin_ctzll(tied);
                tied &= tied - 1;
                const uint32_t key = (uint32_t)__builtin_amdgcn_readlane((int)best_key, l);
                if (key < kmin)
                {
                    kmin = key;
                    win  = l;
                }
            }
            alpha    = wmin;
            blk_obj  = (int)(kmin >> 16);
            blk_ctr  = (uint32_t)__builtin_amdgcn_readlane((int)best_ctr, win);
            blk_type = (uint32_t)__builtin_amdgcn_readlane((int)best_type, win);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        asm volatile("" ::: "memory");
        // the step itself (lexlsi.h:1236-1240, objective.h:585-589); with alpha == 0 the state only moves to its home
        const bool move = alpha > 0.0;
        for (uint32_t j = lane; j < n; j += 64)
        {
            const double xo = src[j];
            st[j]           = move ? xo + alpha * dx_s[j] : xo;
        }
        for (uint32_t g = lane; g < total; g += 64)
        {
            const double v = src[n + g], ax = src[n + total + g];
            st[n + g]         = move ? v + alpha * dv_s[g] : v;
            st[n + total + g] = move ? ax + alpha * adx_s[g] : ax;
        }
    }

    __global__ __launch_bounds__(256) void lsi_step_kernel(StepArgs a)
    {
        extern __shared__ double smem[];
        const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
        const uint32_t b = blockIdx.x * 4 + wib;
        if (b >= a.B) return;
        const uint32_t md = a.mode[b];
        if (md == 0) return;
        const StepShape &sh = a.sh;
        double *dx_s  = smem + (size_t)wib * sh.SD; // n
        double *adx_s = dx_s + sh.n;                // total
        double *dv_s  = adx_s + sh.total;           // total
        double *st         = a.state + (size_t)b * sh.SD;
        const double *src  = (md == 2 ? a.state_in : a.state) + (size_t)b * sh.SD;
        double alpha;
        int blk_obj;
        uint32_t blk_ctr, blk_type;
        lsi_step_wave(sh, a.cdata + (size_t)b * sh.per_data, a.var + (size_t)b * sh.dim0, a.x_lse + (size_t)b * sh.n, src, st, a.ctr_state + (size_t)b * sh.total,
                      a.inact_pos + (size_t)b * sh.total, dx_s, adx_s, dv_s, alpha, blk_obj, blk_ctr, blk_type);
        if (lane == 0)
        {
            double *r = a.res + (size_t)b * 4;
            r[0]      = alpha;
            r[1]      = (double)blk_obj;
            r[2]      = (double)blk_ctr;
            r[3]      = (double)blk_type;
        }
    }

    // =============================================================================================
    // Resident iterations: a whole active-set iteration without the host.  Once an instance has left phase 1 an iteration is
    //   solve the equality problem -> step + ratio test -> ONE working-set change (add the blocking constraint, or remove the one the
    //   removal search names, or stop) -> form the next equality problem
    // (verifyWorkingSet, lexlsi.h:1144-1265).  The first and the last part are this kernel: behind the l-QR kernel and the speculative
    // removal sweep of the stage it runs the step (lsi_step_wave), applies the working-set rules of workingset.h:79-118 (swap-with-last
    // in the inactive list, ordered erase in the active list: they decide ties of later ratio tests and the row order of later
    // equality problems, i.e. index parity), keeps the counters of lexlsi.h:1250-1264 and writes the next problem's dimensions, fixed
    // variables, constraint types and row references (Objective::formLexLSE, objective.h:434-494) straight into the equality solver's
    // in slab, where the next stage's row gather finds them.  The host only enqueues stages and polls the count of finished instances.
    // One wavefront per instance.
    // =============================================================================================
    struct ResidentArgs
    {
        StepShape sh;
        uint32_t B, cap, nObjL, off; // off = 1: objective 0 is the simple-bounds objective (its active bounds are the fixed variables)
        int32_t max_factorizations;
        const double *cdata;
        const uint32_t *var;
        const double *x_lse;       // B x n
        const uint32_t *totalrank; // B
        const int32_t *sens;       // B x 3: found, index in the active list, LexLSE level (-1: a fixed variable)
        double *state;             // B x SD
        uint8_t *ctr_state;        // B x total: activation type per constraint (0 = inactive)
        uint16_t *act, *inact, *inact_pos; // B x total each, objective k in [first[k], first[k] + dim[k])
        uint16_t *na;              // B x STEP_MAX_OBJ: active constraints per objective
        int32_t *info;             // B x 8: status, iterations, activations, deactivations, factorizations, total rank, -, -
        uint8_t *alive;            // B
        uint32_t *finished;        // one counter for the group
        // the equality solver's in slab (lexls_lse_round_layout)
        uint32_t *dims, *nfixed, *fixed_idx;
        double *fixed_val;
        uint8_t *skip;
        int32_t *objidx;
        uint32_t *row_src, *row_ld;
        uint8_t *fixed_type, *ctr_type;
        int32_t *resume; // B: the LexLSE level this iteration's working-set change sits in = the levels the next factorization may read back (NULL: off)
    };

    /// LDS of one instance's wavefront: [dx n | A dx total | dv total] doubles, u16 na[STEP_MAX_OBJ], then the working-set lists
    /// u16 [act | inact | inact_pos] and u8 [ctr_state]
    __host__ __device__ inline size_t resident_lds_per_wave(uint32_t SD, uint32_t total) { return (8 * (size_t)SD + 7 * (size_t)total + 2 * STEP_MAX_OBJ + 15) & ~size_t(15); }

    /// LDS of one instance's wavefront and its slices of the resident arrays
    struct ResidentView
    {
        double *dx_s, *adx_s, *dv_s;
        uint16_t *na, *act, *ina, *ipos;
        uint8_t *cs;
        double *st;
        const double *data;
        const uint32_t *var;
        uint8_t *g_cs;
        uint16_t *g_act, *g_ina, *g_ipos, *g_na;
        int32_t *info;
    };
    __device__ __forceinline__ ResidentView resident_view(const ResidentArgs &a, const uint32_t b, const uint32_t wib)
    {
        extern __shared__ double smem[];
        const StepShape &sh = a.sh;
        const uint32_t n = sh.n, total = sh.total;
        ResidentView v;
        char *wl = reinterpret_cast<char *>(smem) + (size_t)wib * resident_lds_per_wave(sh.SD, total);
        v.dx_s   = reinterpret_cast<double *>(wl);
        v.adx_s  = v.dx_s + n;
        v.dv_s   = v.adx_s + total;
        v.na     = reinterpret_cast<uint16_t *>(v.dv_s + total); // the working sets are edited in LDS and written back
        v.act    = v.na + STEP_MAX_OBJ;
        v.ina    = v.act + total;
        v.ipos   = v.ina + total;
        v.cs     = reinterpret_cast<uint8_t *>(v.ipos + total);
        v.st     = a.state + (size_t)b * sh.SD;
        v.data   = a.cdata + (size_t)b * sh.per_data;
        v.var    = a.var + (size_t)b * sh.dim0;
        v.g_cs   = a.ctr_state + (size_t)b * total;
        v.g_act  = a.act + (size_t)b * total;
        v.g_ina  = a.inact + (size_t)b * total;
        v.g_ipos = a.inact_pos + (size_t)b * total;
        v.g_na   = a.na + (size_t)b * STEP_MAX_OBJ;
        v.info   = a.info + (size_t)b * 8;
        return v;
    }
    __device__ __forceinline__ void resident_load_lists(const ResidentArgs &a, const ResidentView &v, const uint32_t lane)
    {
        for (uint32_t g = lane; g < a.sh.total; g += 64)
        {
            v.act[g]  = v.g_act[g];
            v.ina[g]  = v.g_ina[g];
            v.ipos[g] = v.g_ipos[g];
            v.cs[g]   = v.g_cs[g];
        }
        if (lane < STEP_MAX_OBJ) v.na[lane] = v.g_na[lane];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        asm volatile("" ::: "memory");
    }

    /// what the step of an iteration found (every lane holds it)
    struct StepVerdict
    {
        double alpha;
        int blk_obj;
        uint32_t blk_ctr, blk_type;
    };

    /// first half of a resident iteration: the step and the ratio test on the solution of the equality problem (lsi_step_wave); the state moves.
    /// The instance must be alive.
    __device__ __forceinline__ StepVerdict lsi_iterate_step(const ResidentArgs &a, const uint32_t b, const uint32_t wib)
    {
        const uint32_t lane   = threadIdx.x & 63u;
        const ResidentView v  = resident_view(a, b, wib);
        resident_load_lists(a, v, lane);
        StepVerdict r;
        lsi_step_wave(a.sh, v.data, v.var, a.x_lse + (size_t)b * a.sh.n, v.st, v.st, v.cs, v.ipos, v.dx_s, v.adx_s, v.dv_s, r.alpha, r.blk_obj, r.blk_ctr, r.blk_type);
        return r;
    }

    /// second half: ONE working-set change (the blocking constraint joins, or the constraint the removal search named leaves, or the instance
    /// stops), the counters, the next equality problem.  The removal search's verdict (a.sens) is read only when the step was not blocked —
    /// a caller that runs the search between the two halves may skip it for a blocked step, as the reference does (lexlsi.h:1181-1232).
    /// RELOAD: the working-set lists are not in LDS any more (something else used it since lsi_iterate_step)
    template <bool RELOAD>
    __device__ __forceinline__ void lsi_iterate_finish(const ResidentArgs &a, const uint32_t b, const uint32_t wib, const StepVerdict &verdict)
    {
        const uint32_t lane  = threadIdx.x & 63u;
        const StepShape &sh  = a.sh;
        const uint32_t n = sh.n, total = sh.total;
        const ResidentView v = resident_view(a, b, wib);
        if (RELOAD) resident_load_lists(a, v, lane);
        uint16_t *na = v.na, *act = v.act, *ina = v.ina, *ipos = v.ipos;
        uint8_t *cs  = v.cs;
        const double *data  = v.data;
        const uint32_t *var = v.var;
        int32_t *info       = v.info;
        const int blk_obj       = verdict.blk_obj;
        const uint32_t blk_ctr  = verdict.blk_ctr, blk_type = verdict.blk_type;
        const bool blocked      = blk_obj >= 0;
        const int32_t found_i = blocked ? 0 : a.sens[(size_t)b * 3], rm_pos = blocked ? -1 : a.sens[(size_t)b * 3 + 1], rm_lvl = blocked ? -2 : a.sens[(size_t)b * 3 + 2];
        const int32_t nfact   = info[4] + 1; // lexlsi.h:1172
        const int32_t niter = info[1], nact = info[2], ndeact = info[3];
        const uint32_t trank = a.totalrank[b];
        (void)n;

        // ---- one working-set change (lexlsi.h:1181-1232) and the counters; lane 0 on the LDS copy ----
        const bool removed = !blocked && found_i != 0;
        const bool done    = (!blocked && !removed) || nfact >= a.max_factorizations; // lexlsi.h:236-240
        if (lane == 0)
        {
            if (blocked) // OPERATION_ADD: workingset.h:79-92
            {
                const uint32_t f = sh.first[blk_obj], nak = na[blk_obj], nik = sh.dim[blk_obj] - nak;
                const uint32_t pos = ipos[f + blk_ctr], last = ina[f + nik - 1];
                ina[f + pos]       = (uint16_t)last;
                ipos[f + last]     = (uint16_t)pos;
                cs[f + blk_ctr]    = (uint8_t)blk_type;
                act[f + nak]       = (uint16_t)blk_ctr;
                na[blk_obj]        = (uint16_t)(nak + 1);
            }
            else if (removed) // OPERATION_REMOVE: workingset.h:99-108
            {
                const uint32_t k = (uint32_t)(rm_lvl + (int32_t)a.off), p = (uint32_t)rm_pos;
                const uint32_t f = sh.first[k], nak = na[k], nik = sh.dim[k] - nak;
                const uint32_t c = act[f + p];
                for (uint32_t i = p; i + 1 < nak; i++) act[f + i] = act[f + i + 1];
                cs[f + c]    = (uint8_t)CTR_INACTIVE;
                ina[f + nik] = (uint16_t)c;
                ipos[f + c]  = (uint16_t)nik;
                na[k]        = (uint16_t)(nak - 1);
            }
            // prefix reuse: the change touches ONE objective; the equality problem's levels above it keep their rows (an activation appends to
            // its level, a removal erases in order: workingset.h:79-108), so the next factorization reads them back.  A change in the
            // simple-bounds objective changes the fixed variables: everything again.
            if (a.resume)
            {
                const int32_t K = blocked ? (blk_obj >= (int)a.off ? blk_obj - (int)a.off : 0) : (removed && rm_lvl > 0 ? rm_lvl : 0);
                a.resume[b]     = K;
                if (!done) info[6] += K, info[7] += 1;
            }
            info[0] = (!blocked && !removed) ? (int32_t)PROBLEM_SOLVED : (done ? (int32_t)MAX_NUMBER_OF_FACTORIZATIONS_EXCEEDED : info[0]);
            info[1] = niter + 1;
            info[2] = nact + (blocked ? 1 : 0);
            info[3] = ndeact + (removed ? 1 : 0);
            info[4] = nfact;
            info[5] = (int32_t)trank;
            if (done)
            {
                a.alive[b]  = 0;
                a.skip[b]   = 1;
                a.objidx[b] = -1;
                atomicAdd(a.finished, 1u);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        asm volatile("" ::: "memory");
        for (uint32_t g = lane; g < total; g += 64) // (a change touches a handful of entries; the lists are short: all of them go back)
        {
            v.g_act[g]  = act[g];
            v.g_ina[g]  = ina[g];
            v.g_ipos[g] = ipos[g];
            v.g_cs[g]   = cs[g];
        }
        if (lane < STEP_MAX_OBJ) v.g_na[lane] = na[lane];
        if (done) return;

        // ---- the next equality problem (lexlsi.h:968-982, objective.h:434-494), lane = active constraint ----
        uint32_t counter = 0;
        for (uint32_t k = 0; k < sh.nObj; k++)
        {
            const uint32_t f = sh.first[k], dim = sh.dim[k], nak = na[k];
            const double *blk = data + sh.off[k];
            if (sh.simple[k])
            {
                if (lane == 0) a.nfixed[b] = nak;
                for (uint32_t i = lane; i < nak; i += 64)
                {
                    const uint32_t c = act[f + i], t = cs[f + c];
                    const size_t o   = (size_t)b * n + i;
                    a.fixed_idx[o]   = var[c];
                    a.fixed_val[o]   = (t == CTR_ACTIVE_LB) ? blk[c] : blk[c + dim];
                    a.fixed_type[o]  = (uint8_t)t;
                }
            }
            else
            {
                if (lane == 0) a.dims[(size_t)b * a.nObjL + k - a.off] = nak;
                for (uint32_t i = lane; i < nak; i += 64)
                {
                    const uint32_t c = act[f + i], t = cs[f + c];
                    const size_t o   = (size_t)b * a.cap + counter + i;
                    a.row_src[o]     = (uint32_t)(sh.off[k] + c);
                    a.row_ld[o]      = dim | (t == CTR_ACTIVE_LB ? 0u : 0x80000000u);
                    a.ctr_type[o]    = (uint8_t)t;
                }
                counter += nak;
            }
        }
        for (uint32_t r = counter + lane; r < a.cap; r += 64) a.row_ld[(size_t)b * a.cap + r] = 0u;
    }

    /// one instance's wavefront: b = the instance, wib = the wavefront's slice of the dynamic LDS
    __device__ __forceinline__ void lsi_iterate_body(const ResidentArgs &a, const uint32_t b, const uint32_t wib)
    {
        if (!a.alive[b]) return;
        const StepVerdict verdict = lsi_iterate_step(a, b, wib);
        lsi_iterate_finish<false>(a, b, wib, verdict);
    }

    __global__ __launch_bounds__(256) void lsi_iterate_kernel(ResidentArgs a)
    {
        const uint32_t wib = threadIdx.x >> 6;
        const uint32_t b   = blockIdx.x * 4 + wib;
        if (b >= a.B) return;
        lsi_iterate_body(a, b, wib);
    }
} // namespace
