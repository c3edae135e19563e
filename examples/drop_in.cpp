// What a user of the reference writes, compiled against THIS repository's include/ and linked with liblexls_hip.so
// instead of Eigen: the LexLSI active-set driver stays on the host, every factorize/solve/ObjectiveSensitivity of the
// equality problems runs on the GPU through the C ABI (include/lexls_hip.h).
//
//   g++ -std=c++14 -O2 -Iinclude examples/drop_in.cpp -Llexls_amd/csrc -llexls_hip -Wl,-rpath,$PWD/lexls_amd/csrc -o drop_in
//   ./drop_in tests/golden/test_01.dat
//
// Mirrors the call sequence of the reference's MEX front end (interfaces/matlab-octave/lexlsi.cpp:527-625) and of its
// tests/test_01.cpp (which parses the same file); prints the distance to the file's #Solution block.
#include <lexls/lexls.h>
#include <lexls/tools.h>

#include <cmath>
#include <cstdio>
#include <vector>

int main(int argc, char **argv)
{
    if (argc != 2)
    {
        std::fprintf(stderr, "usage: %s hierarchy.dat\n", argv[0]);
        return 2;
    }
    try
    {
        LexLS::tools::Hierarchy h;
        LexLS::tools::HierarchyFileProcessor().import(argv[1], h);

        // ---- inequality hierarchy through LexLSI ----
        std::vector<LexLS::Index> dims(h.number_of_constraints.begin(), h.number_of_constraints.end());
        std::vector<LexLS::ObjectiveType> types(h.types_of_objectives.begin(), h.types_of_objectives.end());
        LexLS::internal::LexLSI lsi(h.number_of_variables, h.number_of_objectives, dims.data(), types.data());
        std::vector<LexLS::Index> var_index = LexLS::tools::HierarchyFileProcessor::simple_bound_indices(h, true);
        for (LexLS::Index k = 0; k < h.number_of_objectives; k++)
        {
            const LexLS::dMatrixType &m = h.objectives[k];
            if (types[k] == LexLS::SIMPLE_BOUNDS_OBJECTIVE)
            {
                // the file stores [index lb ub]; the solver takes the indices separately and [lb ub] as data
                LexLS::dMatrixType bounds(m.rows(), 2);
                for (LexLS::Index i = 0; i < m.rows(); i++)
                {
                    bounds(i, 0) = m(i, 1);
                    bounds(i, 1) = m(i, 2);
                }
                lsi.setData(k, var_index.data(), LexLS::dMatrixConstRef(bounds.data(), bounds.rows(), 2));
            }
            else
                lsi.setData(k, LexLS::dMatrixConstRef(m.data(), m.rows(), m.cols()));
        }
        const LexLS::TerminationStatus status = lsi.solve();
        const LexLS::dVectorType &x           = lsi.get_x();
        double err                            = 0.0;
        for (LexLS::Index i = 0; i < h.solution.size(); i++) err = std::fmax(err, std::fabs(x(i) - h.solution(i)));
        std::printf("LexLSI: status %d, %u factorizations, %u activations, %u deactivations, max|x - #Solution| = %.3e\n", (int)status,
                    (unsigned)lsi.getFactorizationsCount(), (unsigned)lsi.getActivationsCount(), (unsigned)lsi.getDeactivationsCount(), err);

        // ---- one equality problem through LexLSE (the class LexLSI drives), as test_numerical_error.cpp:93-130 does ----
        LexLS::Index edims[2] = {3, 2};
        LexLS::internal::LexLSE lse(4, 2, edims);
        const double A0[3 * 5] = {1, 0, 2, /**/ 0, 1, 1, /**/ 1, 1, 0, /**/ 0, 2, 1, /**/ 1, 2, 3}; // column-major [A | b], 3 x (4+1)
        const double A1[2 * 5] = {1, 0, /**/ 0, 1, /**/ 1, 1, /**/ 2, 0, /**/ 1, -1};
        lse.setData(0, LexLS::dMatrixConstRef(A0, 3, 5));
        lse.setData(1, LexLS::dMatrixConstRef(A1, 2, 5));
        lse.factorize();
        lse.solve();
        const LexLS::dVectorType &xe = lse.get_x();
        const LexLS::dVectorType &v  = lse.get_v();
        std::printf("LexLSE: ranks %u %u, x = [%.6f %.6f %.6f %.6f], |v| of level 0 = %.3e\n", (unsigned)lse.getRank(0), (unsigned)lse.getRank(1), xe(0),
                    xe(1), xe(2), xe(3), std::sqrt(v(0) * v(0) + v(1) * v(1) + v(2) * v(2)));
        // the multipliers of both objectives: the one-argument overload (lexlse.h:770-861) against the deciding one with zero tolerances
        bool lambda_ok = true;
        for (LexLS::Index k = 0; k < 2; k++)
        {
            lse.ObjectiveSensitivity(k);
            const LexLS::dVectorType la = lse.getWorkspace();
            LexLS::Index ctr = 0;
            int obj          = 0;
            LexLS::RealScalar worst = 0;
            lse.ObjectiveSensitivity(k, ctr, obj, 0.0, 0.0, worst);
            const LexLS::dVectorType &lb = lse.getWorkspace();
            const LexLS::Index nl        = k == 0 ? 3 : 5;
            for (LexLS::Index i = 0; i < nl; i++) lambda_ok = lambda_ok && la(i) == lb(i);
        }
        std::printf("LexLSE: multipliers of the one-argument overload %s\n", lambda_ok ? "agree" : "DIFFER");
        return (status == LexLS::PROBLEM_SOLVED && err < 1e-9 && lambda_ok) ? 0 : 1;
    }
    catch (const std::exception &e)
    {
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 3;
    }
}
