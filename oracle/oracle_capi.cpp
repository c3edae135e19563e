// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/lexlse_oracle.h).  Flat C entry points over the
// CPU restatement so that tests/, smoke() and bench.py's cpu_baseline leg can drive it via ctypes.
// Nothing in the product links or loads this library.
#include "lexlse_oracle.h"

#include <chrono>
#include <cstdint>
#include <lexls/lsi_runner.h>
#include <atomic>
#include <thread>

using namespace LexLS;
typedef lexls_oracle::LexLSE OLSE;
typedef internal::LexLSI_T<OLSE> OLSI;

static thread_local std::string g_err;

namespace
{
    struct LseBatchArgs
    {
        uint32_t batch, nVar, nObj;
        const uint32_t *maxdim;   // nObj capacities
        const uint32_t *dims;     // batch x nObj
        const double *lod;        // batch x cap x (nVar+1), column-major, ld = cap
        double tol;
        const uint32_t *nfixed;   // batch or NULL
        const uint32_t *fixed_idx; // batch x nVar
        const double *fixed_val;   // batch x nVar
        const uint8_t *fixed_type; // batch x nVar
        const uint8_t *ctr_type;   // batch x cap or NULL
        int solve_option;          // 0 basic solution, 1 least-norm (Givens), 2 least-norm (normal equations), <0 factorize only
        int sens_obj;              // -1: none, else the level passed to ObjectiveSensitivity
        double tol_wrong, tol_correct;
        // outputs (any may be NULL)
        double *x;          // batch x nVar
        double *factor;     // batch x cap x (nVar+1)
        double *hh;         // batch x cap
        uint32_t *perm;     // batch x nVar
        uint32_t *rank;     // batch x nObj
        uint32_t *fcol;     // batch x nObj
        uint32_t *totalrank; // batch
        double *v;          // batch x cap
        double *lambda;     // batch x (nVar + cap)   [lambda_fixed; lambda]
        int32_t *sens;      // batch x 3: found, CtrIndex2Remove, ObjIndex2Remove
        double *maxabs;     // batch
        uint8_t *ctr_type_out; // batch x cap
    };

    /// regularization of the next oracle_lse_run calls (oracle_lse_set_regularization): type, per-level factors, variable factor
    struct RegState
    {
        int type = 0;
        std::vector<double> factors;
        double variable = 0.0;
        unsigned cg_iterations = 10; // typedefs.h:170
        double *x_mu = NULL, *x_mu_rhs = NULL, *residual_mu = NULL; // by-products of type 7 (batch x nObj x nVar, same, batch x cap)
    } g_reg;

    void run_one(const LseBatchArgs &a, uint32_t b, OLSE &lse, uint32_t cap)
    {
        const uint32_t n = a.nVar;
        std::vector<Index> dims(a.dims + static_cast<size_t>(b) * a.nObj, a.dims + static_cast<size_t>(b + 1) * a.nObj);
        if (a.nfixed && a.nfixed[b] > 0) lse.setFixedVariablesCount(a.nfixed[b]);
        else lse.setFixedVariablesCount(0);
        lse.setObjDim(dims.data());
        if (a.nfixed)
            for (uint32_t k = 0; k < a.nfixed[b]; k++)
                lse.fixVariable(a.fixed_idx[static_cast<size_t>(b) * n + k], a.fixed_val[static_cast<size_t>(b) * n + k],
                                a.fixed_type ? static_cast<ConstraintActivationType>(a.fixed_type[static_cast<size_t>(b) * n + k]) : CTR_ACTIVE_UB);
        lse.setProblem(dMatrixConstRef(a.lod + static_cast<size_t>(b) * cap * (n + 1), cap, n + 1));
        for (uint32_t k = 0; k < a.nObj; k++) lse.setRegularizationFactor(k, k < g_reg.factors.size() ? g_reg.factors[k] : 0.0);
        if (a.ctr_type)
        {
            uint32_t r = 0;
            for (uint32_t k = 0; k < a.nObj; k++)
                for (uint32_t j = 0; j < dims[k]; j++, r++) lse.setCtrType(k, j, static_cast<ConstraintActivationType>(a.ctr_type[static_cast<size_t>(b) * cap + r]));
        }
        lse.factorize();
        if (a.solve_option == 0) lse.solve();
        else if (a.solve_option == 1) lse.solveLeastNorm_1();
        else if (a.solve_option == 2) lse.solveLeastNorm_2();
        else if (a.solve_option == 3) lse.solveLeastNorm_3();

        if (a.x)
            for (uint32_t i = 0; i < n; i++) a.x[static_cast<size_t>(b) * n + i] = lse.get_x()(i);
        if (a.factor)
        {
            const dMatrixType &f = lse.get_lexqr();
            std::memcpy(a.factor + static_cast<size_t>(b) * cap * (n + 1), f.data(), sizeof(double) * cap * (n + 1));
        }
        if (a.hh)
            for (uint32_t i = 0; i < cap; i++) a.hh[static_cast<size_t>(b) * cap + i] = lse.get_hh_scalars()(i);
        if (a.perm)
            for (uint32_t i = 0; i < n; i++) a.perm[static_cast<size_t>(b) * n + i] = i < lse.getTotalRank() ? lse.get_column_permutations()(i) : i;
        for (uint32_t k = 0; k < a.nObj; k++)
        {
            if (a.rank) a.rank[static_cast<size_t>(b) * a.nObj + k] = lse.getRank(k);
            if (a.fcol) a.fcol[static_cast<size_t>(b) * a.nObj + k] = lse.getFirstColIndex(k);
        }
        if (a.totalrank) a.totalrank[b] = lse.getTotalRank();
        if (a.v)
        {
            const dVectorType &w = lse.get_v();
            for (uint32_t i = 0; i < cap; i++) a.v[static_cast<size_t>(b) * cap + i] = i < lse.get_nCtr() ? w(i) : 0.0;
        }
        if (a.sens_obj >= 0)
        {
            Index ctr = 0;
            int obj   = -2;
            double m  = 0;
            const bool found = lse.ObjectiveSensitivity(static_cast<Index>(a.sens_obj), ctr, obj, a.tol_wrong, a.tol_correct, m);
            if (a.sens)
            {
                a.sens[static_cast<size_t>(b) * 3 + 0] = found ? 1 : 0;
                a.sens[static_cast<size_t>(b) * 3 + 1] = found ? static_cast<int32_t>(ctr) : -1;
                a.sens[static_cast<size_t>(b) * 3 + 2] = found ? obj : -2;
            }
            if (a.maxabs) a.maxabs[b] = m;
            if (a.lambda)
            {
                Index nLambda = 0;
                for (int k = 0; k <= a.sens_obj; k++) nLambda += dims[k];
                const Index nf = lse.getFixedVariablesCount();
                double *out    = a.lambda + static_cast<size_t>(b) * (n + cap);
                for (uint32_t i = 0; i < n + cap; i++) out[i] = 0.0;
                for (Index i = 0; i < nf + nLambda; i++) out[i] = lse.getWorkspace()(i);
            }
            if (a.ctr_type_out)
                for (uint32_t i = 0; i < cap; i++) a.ctr_type_out[static_cast<size_t>(b) * cap + i] = static_cast<uint8_t>(lse.get_ctr_type()[i]);
        }
        for (uint32_t k = 0; k < a.nObj; k++) // after the sensitivity call: its initialize_rhs() fills a column of X_mu_rhs
            for (uint32_t i = 0; i < n; i++)
            {
                if (g_reg.x_mu) g_reg.x_mu[(static_cast<size_t>(b) * a.nObj + k) * n + i] = lse.get_X_mu()(i, k);
                if (g_reg.x_mu_rhs) g_reg.x_mu_rhs[(static_cast<size_t>(b) * a.nObj + k) * n + i] = lse.get_X_mu_rhs()(i, k);
            }
        if (g_reg.residual_mu)
            for (uint32_t i = 0; i < cap; i++) g_reg.residual_mu[static_cast<size_t>(b) * cap + i] = lse.get_residual_mu()(i);
    }

    int run_range(const LseBatchArgs &a, uint32_t b0, uint32_t b1)
    {
        std::vector<Index> maxdim(a.maxdim, a.maxdim + a.nObj);
        uint32_t cap = 0;
        for (uint32_t k = 0; k < a.nObj; k++) cap += maxdim[k];
        OLSE lse;
        lse.resize(a.nVar, a.nObj, maxdim.data());
        ParametersLexLSE p;
        p.tol_linear_dependence = a.tol;
        p.regularization_type            = static_cast<RegularizationType>(g_reg.type);
        p.variable_regularization_factor = g_reg.variable;
        p.max_number_of_CG_iterations    = g_reg.cg_iterations;
        lse.setParameters(p);
        for (uint32_t b = b0; b < b1; b++) run_one(a, b, lse, cap);
        return 0;
    }
} // namespace

extern "C"
{
    const char *oracle_last_error() { return g_err.c_str(); }

    /// regularization used by the following oracle_lse_run calls: type = LexLS::RegularizationType, factors = one per level (or NULL)
    void oracle_lse_set_regularization(int type, uint32_t nObj, const double *factors, double variable_factor, uint32_t cg_iterations)
    {
        g_reg.cg_iterations = cg_iterations ? cg_iterations : 10;
        g_reg.type = type;
        g_reg.factors.assign(factors ? factors : NULL, factors ? factors + nObj : NULL);
        g_reg.variable = variable_factor;
    }

    /// where the following oracle_lse_run calls leave X_mu / X_mu_rhs (batch x nObj x nVar: column k of problem b is contiguous) and
    /// residual_mu (batch x cap), lexlse.h:1636-1650; NULL = not wanted
    void oracle_lse_set_mu_outputs(double *x_mu, double *x_mu_rhs, double *residual_mu)
    {
        g_reg.x_mu        = x_mu;
        g_reg.x_mu_rhs    = x_mu_rhs;
        g_reg.residual_mu = residual_mu;
    }

    /// factorize (+ solve, + residual, + sensitivity) of a batch of equality problems, `nthreads` host threads
    int oracle_lse_run(uint32_t batch, uint32_t nVar, uint32_t nObj, const uint32_t *maxdim, const uint32_t *dims, const double *lod, double tol,
                       const uint32_t *nfixed, const uint32_t *fixed_idx, const double *fixed_val, const uint8_t *fixed_type, const uint8_t *ctr_type,
                       int solve_option, int sens_obj, double tol_wrong, double tol_correct, double *x, double *factor, double *hh, uint32_t *perm,
                       uint32_t *rank, uint32_t *fcol, uint32_t *totalrank, double *v, double *lambda, int32_t *sens, double *maxabs, uint8_t *ctr_type_out,
                       int nthreads)
    {
        try
        {
            LseBatchArgs a = {batch, nVar, nObj, maxdim, dims, lod, tol, nfixed, fixed_idx, fixed_val, fixed_type, ctr_type, solve_option, sens_obj,
                              tol_wrong, tol_correct, x, factor, hh, perm, rank, fcol, totalrank, v, lambda, sens, maxabs, ctr_type_out};
            if (nthreads <= 1 || batch < 2) return run_range(a, 0, batch);
            std::vector<std::thread> th;
            std::vector<int> rc(nthreads, 0);
            for (int t = 0; t < nthreads; t++)
            {
                const uint32_t b0 = static_cast<uint32_t>(static_cast<uint64_t>(batch) * t / nthreads);
                const uint32_t b1 = static_cast<uint32_t>(static_cast<uint64_t>(batch) * (t + 1) / nthreads);
                th.emplace_back([&a, &rc, t, b0, b1]() {
                    try { rc[t] = run_range(a, b0, b1); }
                    catch (...) { rc[t] = 1; }
                });
            }
            for (auto &t : th) t.join();
            for (int t = 0; t < nthreads; t++)
                if (rc[t]) return rc[t];
            return 0;
        }
        catch (const std::exception &e)
        {
            g_err = e.what();
            return 1;
        }
    }

    /// wall-clock seconds for `repeats` passes of factorize+solve over the batch (x written to `x`).  The threads are started ONCE: each
    /// allocates its solver once and loops over its block of problems `repeats` times (a pass per spawn would time thread creation and
    /// the allocations of resize(), not the factorizations); the clock starts when every thread is ready and stops when the last one is done
    double oracle_lse_time(uint32_t batch, uint32_t nVar, uint32_t nObj, const uint32_t *maxdim, const uint32_t *dims, const double *lod, double tol,
                           double *x, int nthreads, int repeats)
    {
        LseBatchArgs a = {batch, nVar, nObj, maxdim, dims, lod, tol, NULL, NULL, NULL, NULL, NULL, 0, -1, 0, 0, x, NULL, NULL, NULL, NULL, NULL, NULL, NULL,
                          NULL, NULL, NULL, NULL};
        if (nthreads < 1) nthreads = 1;
        if ((uint32_t)nthreads > batch) nthreads = (int)batch;
        std::atomic<int> ready(0);
        std::atomic<bool> go(false);
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; t++)
        {
            const uint32_t b0 = static_cast<uint32_t>(static_cast<uint64_t>(batch) * t / nthreads);
            const uint32_t b1 = static_cast<uint32_t>(static_cast<uint64_t>(batch) * (t + 1) / nthreads);
            th.emplace_back([&a, &ready, &go, b0, b1, repeats]() {
                std::vector<Index> md(a.maxdim, a.maxdim + a.nObj);
                uint32_t cap = 0;
                for (uint32_t k = 0; k < a.nObj; k++) cap += md[k];
                OLSE lse;
                lse.resize(a.nVar, a.nObj, md.data());
                ParametersLexLSE p;
                p.tol_linear_dependence = a.tol;
                lse.setParameters(p);
                ready.fetch_add(1);
                while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
                for (int r = 0; r < repeats; r++)
                    for (uint32_t b = b0; b < b1; b++) run_one(a, b, lse, cap);
            });
        }
        while (ready.load() < nthreads) std::this_thread::yield();
        const auto t0 = std::chrono::steady_clock::now();
        go.store(true, std::memory_order_release);
        for (auto &t : th) t.join();
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }

    int oracle_hardware_threads() { return static_cast<int>(std::thread::hardware_concurrency()); }

    /// `batch` LexLSI problems of one structure (packed as lexls_lsi_batch_run takes them: data batch x per_data, var_index batch x dims[0],
    /// active_guess batch x total, x0 batch x nVar), default parameters, on `nthreads` host threads (instances dealt out dynamically); what comes
    /// back is the count of factorizations and the wall time of the solves — bench.py's CPU figure for configs[4] on the host's cores
    int oracle_lsi_time_batch(uint32_t batch, uint32_t nVar, uint32_t nObj, const uint32_t *dims, const int32_t *types, const double *data, const uint32_t *var_index,
                              const uint8_t *active_guess, const double *x0, int nthreads, int64_t *factorizations, double *seconds)
    {
        try
        {
            uint64_t per_data = 0, total = 0;
            for (uint32_t k = 0; k < nObj; k++)
            {
                per_data += (uint64_t)dims[k] * (types[k] == 1 ? 2 : nVar + 2);
                total += dims[k];
            }
            const uint32_t dim0 = (nObj && types[0] == 1) ? dims[0] : 0;
            std::atomic<uint32_t> next(0);
            std::atomic<int64_t> nf(0);
            std::atomic<bool> failed(false);
            auto work = [&]() {
                std::vector<double> x(nVar), v(total);
                std::vector<uint8_t> act(total);
                for (;;)
                {
                    const uint32_t b = next.fetch_add(1);
                    if (b >= batch) break;
                    try
                    {
                        runner::LsiProblem p = {nVar, nObj, dims, types, data + (size_t)b * per_data, (var_index && dim0) ? var_index + (size_t)b * dim0 : nullptr,
                                                active_guess ? active_guess + (size_t)b * total : nullptr, x0 ? x0 + (size_t)b * nVar : nullptr};
                        ParametersLexLSI par;
                        runner::LsiInfo info;
                        runner::solve<OLSI>(p, par, x.data(), &info, act.data(), v.data());
                        nf.fetch_add(info.factorizations);
                    }
                    catch (...)
                    {
                        failed.store(true);
                    }
                }
            };
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int i = 1; i < nthreads; i++) th.emplace_back(work);
            work();
            for (auto &t : th) t.join();
            *seconds        = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            *factorizations = nf.load();
            if (failed.load()) throw Exception("oracle_lsi_time_batch: an instance failed");
            return 0;
        }
        catch (const std::exception &e)
        {
            g_err = e.what();
            return 1;
        }
    }

    /// LexLSI on flat arrays (layout: include/lexls/lsi_runner.h).  params: [max_fact, tol_lin_dep, tol_wrong, tol_correct, tol_feas,
    /// cycling(0/1), cycling_max, cycling_relax, deactivate_first_wrong_sign(0/1), use_resumable_form(0/1)] or NULL for defaults.
    int oracle_lsi_run(uint32_t nVar, uint32_t nObj, const uint32_t *dims, const int32_t *types, const double *data, const uint32_t *var_index,
                       const uint8_t *active_guess, const double *x0, const double *params, double *x_out, int32_t *info6, uint8_t *active_out, double *v_out)
    {
        try
        {
            runner::LsiProblem p = {nVar, nObj, dims, types, data, var_index, active_guess, x0};
            ParametersLexLSI par;
            if (params)
            {
                par.max_number_of_factorizations = static_cast<Index>(params[0]);
                par.tol_linear_dependence        = params[1];
                par.tol_wrong_sign_lambda        = params[2];
                par.tol_correct_sign_lambda      = params[3];
                par.tol_feasibility              = params[4];
                par.cycling_handling_enabled     = params[5] != 0;
                par.cycling_max_counter          = static_cast<Index>(params[6]);
                par.cycling_relax_step           = params[7];
                par.deactivate_first_wrong_sign  = params[8] != 0;
            }
            runner::LsiInfo info;
            if (params && params[9] != 0) // the resumable form of the driver (what lock-step batches use)
            {
                OLSI lsi;
                runner::setup(lsi, p, par);
                lsi.solve_resumable();
                runner::collect(lsi, p, x_out, &info, active_out, v_out);
            }
            else
            {
                runner::solve<OLSI>(p, par, x_out, &info, active_out, v_out);
            }
            if (info6) std::memcpy(info6, &info, sizeof(info));
            return 0;
        }
        catch (const std::exception &e)
        {
            g_err = e.what();
            return 1;
        }
    }

    /// oracle_lsi_run with initial residuals, regularization factors and the three regularization parameters (params: 12 values)
    int oracle_lsi_run_ex(uint32_t nVar, uint32_t nObj, const uint32_t *dims, const int32_t *types, const double *data, const uint32_t *var_index,
                          const uint8_t *active_guess, const double *x0, const double *v0, const double *reg_factors, const double *params12,
                          double *x_out, int32_t *info6, uint8_t *active_out, double *v_out)
    {
        try
        {
            runner::LsiProblem p = {nVar, nObj, dims, types, data, var_index, active_guess, x0, v0, reg_factors};
            ParametersLexLSI par;
            if (params12)
            {
                par.max_number_of_factorizations   = static_cast<Index>(params12[0]);
                par.tol_linear_dependence          = params12[1];
                par.tol_wrong_sign_lambda          = params12[2];
                par.tol_correct_sign_lambda        = params12[3];
                par.tol_feasibility                = params12[4];
                par.cycling_handling_enabled       = params12[5] != 0;
                par.cycling_max_counter            = static_cast<Index>(params12[6]);
                par.cycling_relax_step             = params12[7];
                par.deactivate_first_wrong_sign    = params12[8] != 0;
                par.regularization_type            = static_cast<RegularizationType>(static_cast<int>(params12[9]));
                par.variable_regularization_factor = params12[10];
                par.max_number_of_CG_iterations    = static_cast<Index>(params12[11]);
            }
            runner::LsiInfo info;
            runner::solve<OLSI>(p, par, x_out, &info, active_out, v_out);
            if (info6) std::memcpy(info6, &info, sizeof(info));
            return 0;
        }
        catch (const std::exception &e)
        {
            g_err = e.what();
            return 1;
        }
    }

    /// oracle_lsi_run_ex + the debug outputs of the MEX front end (runner::collect_debug); `dbg` = 12 pointers / values in the order of
    /// runner::LsiDebug: lambda, lexqr, data, x_star, active_ctr, log, log_alpha, max_log (as uintptr), x_mu, x_mu_rhs, residual_mu, counts
    int oracle_lsi_run_debug(uint32_t nVar, uint32_t nObj, const uint32_t *dims, const int32_t *types, const double *data, const uint32_t *var_index,
                             const uint8_t *active_guess, const double *x0, const double *v0, const double *reg_factors, const double *params12,
                             double *x_out, int32_t *info6, uint8_t *active_out, double *v_out, double *lambda, double *lexqr, double *pdata, double *x_star,
                             int32_t *active_ctr, int32_t *log, double *log_alpha, uint32_t max_log, double *x_mu, double *x_mu_rhs, double *residual_mu,
                             uint32_t *counts)
    {
        try
        {
            runner::LsiProblem p = {nVar, nObj, dims, types, data, var_index, active_guess, x0, v0, reg_factors};
            ParametersLexLSI par;
            if (params12)
            {
                par.max_number_of_factorizations   = static_cast<Index>(params12[0]);
                par.tol_linear_dependence          = params12[1];
                par.tol_wrong_sign_lambda          = params12[2];
                par.tol_correct_sign_lambda        = params12[3];
                par.tol_feasibility                = params12[4];
                par.cycling_handling_enabled       = params12[5] != 0;
                par.cycling_max_counter            = static_cast<Index>(params12[6]);
                par.cycling_relax_step             = params12[7];
                par.deactivate_first_wrong_sign    = params12[8] != 0;
                par.regularization_type            = static_cast<RegularizationType>(static_cast<int>(params12[9]));
                par.variable_regularization_factor = params12[10];
                par.max_number_of_CG_iterations    = static_cast<Index>(params12[11]);
            }
            par.log_working_set_enabled = true;
            OLSI lsi;
            runner::setup(lsi, p, par);
            lsi.solve();
            runner::LsiInfo info;
            runner::collect(lsi, p, x_out, &info, active_out, v_out);
            if (info6) std::memcpy(info6, &info, sizeof(info));
            const runner::LsiDebug d = {lambda, lexqr, pdata, x_star, active_ctr, log, log_alpha, max_log, x_mu, x_mu_rhs, residual_mu, counts};
            runner::collect_debug(lsi, p, par, d);
            return 0;
        }
        catch (const std::exception &e)
        {
            g_err = e.what();
            return 1;
        }
    }

    /// multipliers of a solved-from-scratch LexLSI problem: lambda_out is sum(dims) x nObj, column-major (lexlsi.h:552-605)
    int oracle_lsi_lambda(uint32_t nVar, uint32_t nObj, const uint32_t *dims, const int32_t *types, const double *data, const uint32_t *var_index,
                          double *x_out, double *lambda_out)
    {
        try
        {
            runner::LsiProblem p = {nVar, nObj, dims, types, data, var_index, NULL, NULL};
            OLSI lsi;
            runner::setup(lsi, p, ParametersLexLSI());
            lsi.solve();
            for (uint32_t i = 0; i < nVar; i++) x_out[i] = lsi.get_x()(i);
            std::vector<dMatrixType> L;
            lsi.getLambda(L);
            uint32_t total = 0;
            for (uint32_t k = 0; k < nObj; k++) total += dims[k];
            uint32_t r0 = 0;
            for (uint32_t k = 0; k < nObj; k++)
            {
                for (uint32_t i = 0; i < dims[k]; i++)
                    for (uint32_t j = 0; j < nObj; j++) lambda_out[(r0 + i) + static_cast<size_t>(j) * total] = L[k](i, j);
                r0 += dims[k];
            }
            return 0;
        }
        catch (const std::exception &e)
        {
            g_err = e.what();
            return 1;
        }
    }

    /// header of a .dat file: out4 = [nVar, nObj, type_header, has_solution]
    int oracle_dat_header(const char *path, int32_t *out4)
    {
        try
        {
            tools::Hierarchy h;
            tools::HierarchyFileProcessor().import(path, h);
            out4[0] = static_cast<int32_t>(h.number_of_variables);
            out4[1] = static_cast<int32_t>(h.number_of_objectives);
            out4[2] = static_cast<int32_t>(h.type_header);
            out4[3] = h.solution.size() == h.number_of_variables ? 1 : 0;
            return 0;
        }
        catch (const std::exception &e)
        {
            g_err = e.what();
            return 1;
        }
    }

    /// parse a .dat inequality hierarchy and solve it with the host driver over the CPU restatement
    int oracle_lsi_run_dat(const char *path, int one_based, int use_active_guess, int use_x_guess, double *x_out, int32_t *info6, double *solution_out)
    {
        try
        {
            tools::Hierarchy h;
            tools::HierarchyFileProcessor().import(path, h);
            runner::FlatHierarchy f;
            runner::flatten(h, one_based != 0, use_active_guess != 0, use_x_guess != 0, f);
            runner::LsiInfo info;
            runner::solve<OLSI>(f.problem, ParametersLexLSI(), x_out, &info, NULL, NULL);
            if (info6) std::memcpy(info6, &info, sizeof(info));
            if (solution_out)
                for (Index i = 0; i < h.solution.size(); i++) solution_out[i] = h.solution(i);
            return 0;
        }
        catch (const std::exception &e)
        {
            g_err = e.what();
            return 1;
        }
    }
}
