// Flat-array front end for the host active-set driver: builds a LexLSI_T<LSE> from plain buffers,
// solves it and flattens the results.  Shared by the product C ABI (lexls_amd/csrc) and by the
// oracle's C API (oracle/oracle_capi.cpp) so that both sides of a parity test are driven by the
// same call sequence — the sequence of the reference's own front end
// (interfaces/matlab-octave/lexlsi.cpp:527-625: setParameters, setData per objective, set_x0,
// api_activate per guessed constraint, solve).
#pragma once

#include <cstdint>
#include <lexls/lexlsi.h>
#include <lexls/tools.h>

namespace LexLS
{
    namespace runner
    {
        struct LsiProblem
        {
            Index nVar, nObj;
            const Index *dims;           // nObj
            const int32_t *types;        // nObj: 0 general, 1 simple bounds (objective 0 only)
            const double *data;          // objectives concatenated, each column-major: general dim x (nVar+2) [A lb ub], simple dim x 2 [lb ub]
            const Index *var_index;      // dims[0] 0-based indices when objective 0 is simple bounds, else unused
            const uint8_t *active_guess; // sum(dims) activation flags (0..3) or NULL
            const double *x0;            // nVar or NULL
            const double *v0          = NULL; // sum(dims) initial residuals (lexlsi.cpp:573-587, set_v0 per objective) or NULL
            const double *reg_factors = NULL; // nObj regularization factors (lexlsi.cpp:539, setRegularizationFactor) or NULL
        };

        struct LsiInfo
        {
            int32_t status, iterations, activations, deactivations, factorizations, total_rank;
        };

        inline size_t objective_size(const LsiProblem &p, Index k) { return static_cast<size_t>(p.dims[k]) * (p.types[k] == 1 ? 2 : p.nVar + 2); }

        template <class LSI>
        void setup(LSI &lsi, const LsiProblem &p, const ParametersLexLSI &par)
        {
            std::vector<Index> dims(p.dims, p.dims + p.nObj);
            std::vector<ObjectiveType> types(p.nObj);
            for (Index k = 0; k < p.nObj; k++) types[k] = p.types[k] == 1 ? SIMPLE_BOUNDS_OBJECTIVE : GENERAL_OBJECTIVE;
            lsi.resize(p.nVar, p.nObj, dims.data(), types.data());
            lsi.setParameters(par);

            size_t off = 0;
            for (Index k = 0; k < p.nObj; k++)
            {
                if (types[k] == SIMPLE_BOUNDS_OBJECTIVE)
                {
                    std::vector<Index> vi(p.var_index, p.var_index + dims[k]);
                    lsi.setData(k, vi.data(), dMatrixConstRef(p.data + off, dims[k], 2));
                }
                else
                {
                    lsi.setData(k, dMatrixConstRef(p.data + off, dims[k], p.nVar + 2));
                    lsi.setDataOffset(k, off);
                }
                off += objective_size(p, k);
            }
            if (p.reg_factors)
                for (Index k = 0; k < p.nObj; k++)
                    if (!(types[k] == SIMPLE_BOUNDS_OBJECTIVE && p.reg_factors[k] == 0.0)) lsi.setRegularizationFactor(k, p.reg_factors[k]); // (objective.h:820 warns for simple bounds)
            if (p.x0) lsi.set_x0(dVectorType(p.x0, p.nVar));
            if (p.v0)
            {
                size_t r = 0;
                for (Index k = 0; k < p.nObj; k++)
                {
                    dVectorType vk(p.v0 + r, dims[k]);
                    lsi.set_v0(k, vk);
                    r += dims[k];
                }
            }
            if (p.active_guess)
            {
                size_t r = 0;
                for (Index k = 0; k < p.nObj; k++)
                    for (Index j = 0; j < dims[k]; j++, r++)
                        if (p.active_guess[r] != CTR_INACTIVE) lsi.api_activate(k, j, static_cast<ConstraintActivationType>(p.active_guess[r]));
            }
        }

        /// x_out: nVar; active_out / v_out: sum(dims) (either may be NULL)
        template <class LSI>
        void collect(LSI &lsi, const LsiProblem &p, double *x_out, LsiInfo *info, uint8_t *active_out, double *v_out)
        {
            const dVectorType &x = lsi.get_x();
            for (Index i = 0; i < p.nVar; i++) x_out[i] = x(i);
            if (info)
            {
                info->status         = static_cast<int32_t>(lsi.getStatus());
                info->iterations     = static_cast<int32_t>(lsi.getIterationsCount());
                info->activations    = static_cast<int32_t>(lsi.getActivationsCount());
                info->deactivations  = static_cast<int32_t>(lsi.getDeactivationsCount());
                info->factorizations = static_cast<int32_t>(lsi.getFactorizationsCount());
                info->total_rank     = static_cast<int32_t>(lsi.getTotalRank());
            }
            size_t r = 0;
            for (Index k = 0; k < p.nObj; k++)
            {
                std::vector<ConstraintActivationType> t;
                lsi.getActiveCtr(k, t);
                const dVectorType &v = lsi.get_v(k);
                for (Index j = 0; j < p.dims[k]; j++, r++)
                {
                    if (active_out) active_out[r] = static_cast<uint8_t>(t[j]);
                    if (v_out) v_out[r] = v(j);
                }
            }
        }

        /// Where the fifth output of the MEX front end goes (interfaces/matlab-octave/lexlsi.cpp:739-770, formDebugStructure :77-260).
        /// Every pointer may be NULL.  total = sum(dims); rows = row capacity of the equality solver (total, minus dims[0] when
        /// objective 0 holds simple bounds); nObjL = its number of levels.
        struct LsiDebug
        {
            double *lambda;       // total x nObj, column-major: getLambda() with the objectives stacked in the user's constraint order
            double *lexqr, *data; // rows x (nVar+1), column-major, ld = rows: get_lexqr(), get_data() of the last equality problem
            double *x_star;       // nVar: get_xStar()
            int32_t *active_ctr;  // total x 3: (obj_index, ctr_index, ctr_type) in working-set order (getActiveCtr_order)
            int32_t *log;         // max_log x 5: (obj_index, ctr_index, ctr_type, cycling_detected, rank) per working-set change
            double *log_alpha;    // max_log: alpha_or_lambda
            uint32_t max_log;
            double *x_mu, *x_mu_rhs, *residual_mu; // REGULARIZATION_TIKHONOV_1 only: nObjL x nVar (column k contiguous) twice, rows
            uint32_t *counts;     // 4: rows, nObjL, number of active constraints, number of log entries (may exceed max_log: truncated)
        };

        /// the getter sequence of lexlsi.cpp:752-762, in its order (getLambda and get_xStar re-factorize the last equality problem)
        template <class LSI>
        void collect_debug(LSI &lsi, const LsiProblem &p, const ParametersLexLSI &par, const LsiDebug &d)
        {
            Index total = 0;
            for (Index k = 0; k < p.nObj; k++) total += p.dims[k];
            const std::vector<WorkingSetLogEntry> wlog = lsi.getWorkingSetLog();
            if (d.lambda)
            {
                std::vector<dMatrixType> L;
                lsi.getLambda(L);
                Index r0 = 0;
                for (Index k = 0; k < p.nObj; k++)
                {
                    for (Index i = 0; i < p.dims[k]; i++)
                        for (Index j = 0; j < p.nObj; j++) d.lambda[(r0 + i) + static_cast<size_t>(j) * total] = L[k](i, j);
                    r0 += p.dims[k];
                }
            }
            if (d.x_star)
            {
                const dVectorType &xs = lsi.get_xStar();
                for (Index i = 0; i < p.nVar; i++) d.x_star[i] = xs(i);
            }
            const dMatrixType &F = lsi.get_lexqr();
            const Index rows     = F.rows();
            if (d.lexqr)
                for (Index j = 0; j <= p.nVar; j++)
                    for (Index i = 0; i < rows; i++) d.lexqr[i + static_cast<size_t>(j) * rows] = F(i, j);
            Index nObjL = p.nObj - ((p.nObj > 0 && p.types[0] == 1) ? 1 : 0);
            if (par.regularization_type == REGULARIZATION_TIKHONOV_1)
            {
                const dMatrixType &X = lsi.get_X_mu(), &Xr = lsi.get_X_mu_rhs();
                const dVectorType &rm = lsi.get_residual_mu();
                nObjL = X.cols();
                for (Index k = 0; k < nObjL; k++)
                    for (Index i = 0; i < p.nVar; i++)
                    {
                        if (d.x_mu) d.x_mu[static_cast<size_t>(k) * p.nVar + i] = X(i, k);
                        if (d.x_mu_rhs) d.x_mu_rhs[static_cast<size_t>(k) * p.nVar + i] = Xr(i, k);
                    }
                if (d.residual_mu)
                    for (Index i = 0; i < rows; i++) d.residual_mu[i] = rm(i);
            }
            std::vector<ConstraintIdentifier> act;
            lsi.getActiveCtr_order(act);
            if (d.active_ctr)
                for (size_t i = 0; i < act.size() && i < static_cast<size_t>(total); i++)
                {
                    d.active_ctr[3 * i + 0] = static_cast<int32_t>(act[i].obj_index);
                    d.active_ctr[3 * i + 1] = static_cast<int32_t>(act[i].ctr_index);
                    d.active_ctr[3 * i + 2] = static_cast<int32_t>(act[i].ctr_type);
                }
            if (d.data)
            {
                const dMatrixType &D = lsi.get_data();
                for (Index j = 0; j <= p.nVar; j++)
                    for (Index i = 0; i < rows; i++) d.data[i + static_cast<size_t>(j) * rows] = D(i, j);
            }
            for (size_t i = 0; i < wlog.size() && i < d.max_log; i++)
            {
                if (d.log)
                {
                    d.log[5 * i + 0] = static_cast<int32_t>(wlog[i].obj_index);
                    d.log[5 * i + 1] = static_cast<int32_t>(wlog[i].ctr_index);
                    d.log[5 * i + 2] = static_cast<int32_t>(wlog[i].ctr_type);
                    d.log[5 * i + 3] = wlog[i].cycling_detected ? 1 : 0;
                    d.log[5 * i + 4] = static_cast<int32_t>(wlog[i].rank);
                }
                if (d.log_alpha) d.log_alpha[i] = wlog[i].alpha_or_lambda;
            }
            if (d.counts)
            {
                d.counts[0] = static_cast<uint32_t>(rows);
                d.counts[1] = static_cast<uint32_t>(nObjL);
                d.counts[2] = static_cast<uint32_t>(act.size());
                d.counts[3] = static_cast<uint32_t>(wlog.size());
            }
        }

        template <class LSI>
        void solve(const LsiProblem &p, const ParametersLexLSI &par, double *x_out, LsiInfo *info, uint8_t *active_out, double *v_out)
        {
            LSI lsi;
            setup(lsi, p, par);
            lsi.solve();
            collect(lsi, p, x_out, info, active_out, v_out);
        }

        /// flatten a parsed .dat hierarchy (tools.h) into an LsiProblem; storage keeps the buffers alive
        struct FlatHierarchy
        {
            std::vector<Index> dims, var_index;
            std::vector<int32_t> types;
            std::vector<double> data;
            std::vector<uint8_t> guess;
            LsiProblem problem;
        };

        inline void flatten(const tools::Hierarchy &h, bool one_based_simple_bounds, bool use_active_set_guess, bool use_solution_guess, FlatHierarchy &f)
        {
            if (h.type_of_hierarchy != tools::HIERARCHY_TYPE_INEQUALITY) throw Exception("flatten: inequality hierarchy expected");
            f.dims.assign(h.number_of_constraints.begin(), h.number_of_constraints.end());
            f.types.clear();
            f.data.clear();
            f.guess.clear();
            for (Index k = 0; k < h.number_of_objectives; k++)
            {
                const bool simple = h.types_of_objectives[k] == SIMPLE_BOUNDS_OBJECTIVE;
                f.types.push_back(simple ? 1 : 0);
                const dMatrixType &m = h.objectives[k];
                for (Index j = simple ? 1 : 0; j < m.cols(); j++)
                    for (Index i = 0; i < m.rows(); i++) f.data.push_back(m(i, j));
                if (h.type_header == 210)
                    for (Index i = 0; i < m.rows(); i++) f.guess.push_back(static_cast<uint8_t>(h.active_set_guess[k][i]));
            }
            f.var_index            = tools::HierarchyFileProcessor::simple_bound_indices(h, one_based_simple_bounds);
            f.problem.nVar         = h.number_of_variables;
            f.problem.nObj         = h.number_of_objectives;
            f.problem.dims         = f.dims.data();
            f.problem.types        = f.types.data();
            f.problem.data         = f.data.data();
            f.problem.var_index    = f.var_index.empty() ? NULL : f.var_index.data();
            f.problem.active_guess = (use_active_set_guess && !f.guess.empty()) ? f.guess.data() : NULL;
            f.problem.x0           = (use_solution_guess && h.solution_guess.size() == h.number_of_variables) ? h.solution_guess.data() : NULL;
        }
    } // namespace runner
} // namespace LexLS
