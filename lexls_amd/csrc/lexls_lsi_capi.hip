// C ABI for inequality problems: the host-side active-set driver (include/lexls/lexlsi.h) instantiated over
// the HIP-backed equality solver (include/lexls/lexlse.h).  Host code only; the device work happens inside the
// lexls_lse_* calls the driver issues.
#include <lexls/lexls.h>
#include <lexls/lsi_runner.h>

using namespace LexLS;

namespace
{

    ParametersLexLSI unpack(const double *p)
    {
        ParametersLexLSI par;
        if (p)
        {
            par.max_number_of_factorizations = static_cast<Index>(p[0]);
            par.tol_linear_dependence        = p[1];
            par.tol_wrong_sign_lambda        = p[2];
            par.tol_correct_sign_lambda      = p[3];
            par.tol_feasibility              = p[4];
            par.cycling_handling_enabled     = p[5] != 0;
            par.cycling_max_counter          = static_cast<Index>(p[6]);
            par.cycling_relax_step           = p[7];
            par.deactivate_first_wrong_sign  = p[8] != 0;
        }
        return par;
    }
} // namespace

extern "C"
{
    void lexls_internal_set_error(const char *msg);

    int lexls_lsi_solve(int device, uint32_t nVar, uint32_t nObj, const uint32_t *h_dims, const int32_t *h_types, const double *h_data,
                        const uint32_t *h_var_index, const uint8_t *h_active_guess, const double *h_x0, const double *h_params9, double *h_x,
                        int32_t *h_info6, uint8_t *h_active, double *h_v)
    {
        try
        {
            runner::LsiProblem p = {nVar, nObj, h_dims, h_types, h_data, h_var_index, h_active_guess, h_x0};
            internal::LexLSI lsi;
            lsi.getLexLSE().setDevice(device);
            runner::setup(lsi, p, unpack(h_params9));
            lsi.solve();
            runner::LsiInfo info;
            runner::collect(lsi, p, h_x, &info, h_active, h_v);
            if (h_info6) std::memcpy(h_info6, &info, sizeof(info));
            return LEXLS_OK;
        }
        catch (const std::exception &e)
        {
            lexls_internal_set_error(e.what());
            return LEXLS_ERR_INVALID;
        }
    }

    int lexls_lsi_solve_dat(int device, const char *path, int one_based, int use_active_guess, int use_x_guess, double *h_x, int32_t *h_info6,
                            double *h_solution)
    {
        try
        {
            tools::Hierarchy h;
            tools::HierarchyFileProcessor().import(path, h);
            runner::FlatHierarchy f;
            runner::flatten(h, one_based != 0, use_active_guess != 0, use_x_guess != 0, f);
            internal::LexLSI lsi;
            lsi.getLexLSE().setDevice(device);
            runner::setup(lsi, f.problem, ParametersLexLSI());
            lsi.solve();
            runner::LsiInfo info;
            runner::collect(lsi, f.problem, h_x, &info, NULL, NULL);
            if (h_info6) std::memcpy(h_info6, &info, sizeof(info));
            if (h_solution)
                for (Index i = 0; i < h.solution.size(); i++) h_solution[i] = h.solution(i);
            return LEXLS_OK;
        }
        catch (const std::exception &e)
        {
            lexls_internal_set_error(e.what());
            return LEXLS_ERR_INVALID;
        }
    }
}
