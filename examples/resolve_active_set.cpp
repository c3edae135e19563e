// The property the reference's tests/test_numerical_error.cpp checks (its lines 83-137), through the same classes and compiled
// against THIS repository's include/ + liblexls_hip.so: solve an inequality hierarchy with LexLSI, then hand the equality problem
// of its last iteration (get_data(), the active set in working-set order, the fixed variables of the simple-bounds level) to a
// fresh LexLSE and compare solution and factor.
//
// The reference reports small differences there and attributes them to Eigen's vectorisation (test_numerical_error.cpp:5-22).
// On this path every chain has ONE evaluation order (oracle/lexlse_oracle.h), so the factor must come out IDENTICAL, and x equal
// up to the rounding of the driver's final step x + 1*(x_lse - x).
//
//   g++ -std=c++14 -O2 -Iinclude examples/resolve_active_set.cpp -Llexls_amd/csrc -llexls_hip -Wl,-rpath,$PWD/lexls_amd/csrc -o resolve
//   ./resolve tests/golden/test_01.dat
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
        using namespace LexLS;
        tools::Hierarchy h;
        tools::HierarchyFileProcessor().import(argv[1], h);
        const Index n = h.number_of_variables, nObj = h.number_of_objectives;

        std::vector<Index> dims(h.number_of_constraints.begin(), h.number_of_constraints.end());
        std::vector<ObjectiveType> types(h.types_of_objectives.begin(), h.types_of_objectives.end());
        internal::LexLSI lsi(n, nObj, dims.data(), types.data());
        std::vector<Index> var_index = tools::HierarchyFileProcessor::simple_bound_indices(h, true);
        std::vector<dMatrixType> bounds(nObj);
        for (Index k = 0; k < nObj; k++)
        {
            const dMatrixType &m = h.objectives[k];
            if (types[k] == SIMPLE_BOUNDS_OBJECTIVE)
            {
                bounds[k].resize(m.rows(), 2);
                for (Index i = 0; i < m.rows(); i++)
                {
                    bounds[k](i, 0) = m(i, 1);
                    bounds[k](i, 1) = m(i, 2);
                }
                lsi.setData(k, var_index.data(), dMatrixConstRef(bounds[k].data(), bounds[k].rows(), 2));
            }
            else
                lsi.setData(k, dMatrixConstRef(m.data(), m.rows(), m.cols()));
        }
        if (lsi.solve() != PROBLEM_SOLVED) throw Exception("LexLSI did not report PROBLEM_SOLVED");
        const dVectorType xstar = lsi.get_x();

        // the equality problem of the last iteration, as LexLSI formed it
        const dMatrixType data  = lsi.get_data();
        const dMatrixType lexqr = lsi.get_lexqr();
        std::vector<ConstraintIdentifier> order;
        lsi.getActiveCtr_order(order);

        const Index off = (types[0] == SIMPLE_BOUNDS_OBJECTIVE) ? 1 : 0;
        std::vector<Index> eq_dim(nObj - off);
        Index nCtr = 0;
        for (Index k = off; k < nObj; k++)
        {
            eq_dim[k - off] = lsi.getActiveCtrCount(k);
            nCtr += eq_dim[k - off];
        }

        // a fresh equality solver sized for the inequality problem and used with the active dimensions (the reference's
        // INITIALIZE_INEQUALITY variant, test_numerical_error.cpp:92-97)
        internal::LexLSE lse(n, nObj - off, dims.data() + off);
        lse.setObjDim(eq_dim.data());
        size_t next = 0;
        if (off)
        {
            lse.setFixedVariablesCount(lsi.getActiveCtrCount(0));
            for (; next < order.size() && order[next].obj_index == 0; next++)
            {
                const Index c = order[next].ctr_index;
                const ConstraintActivationType t = order[next].ctr_type;
                lse.fixVariable(var_index[c], bounds[0](c, t == CTR_ACTIVE_LB ? 0 : 1), t);
            }
        }
        Index row = 0;
        for (Index k = 0; k < nObj - off; k++)
        {
            lse.setData(k, dMatrixConstRef(&data(row, 0), eq_dim[k], n + 1, data.rows()));
            for (Index j = 0; j < eq_dim[k]; j++, next++) lse.setCtrType(k, j, order[next].ctr_type);
            row += eq_dim[k];
        }
        lse.factorize();
        lse.solve();

        const dVectorType &x1    = lse.get_x();
        const dMatrixType &data1 = lse.get_data();
        const dMatrixType &qr1   = lse.get_lexqr();
        double ex = 0.0, ed = 0.0, eq = 0.0;
        for (Index i = 0; i < n; i++) ex = std::fmax(ex, std::fabs(x1(i) - xstar(i)));
        for (Index j = 0; j <= n; j++)
            for (Index i = 0; i < nCtr; i++)
            {
                ed = std::fmax(ed, std::fabs(data(i, j) - data1(i, j)));
                eq = std::fmax(eq, std::fabs(lexqr(i, j) - qr1(i, j)));
            }
        std::printf("active constraints %u (+%u fixed variables), error(x) = %.3e, error(data) = %.3e, error(lexqr) = %.3e\n", (unsigned)nCtr,
                    (unsigned)(off ? lsi.getActiveCtrCount(0) : 0), ex, ed, eq);
        return (ex <= 1e-12 && ed == 0.0 && eq == 0.0) ? 0 : 1;
    }
    catch (const std::exception &e)
    {
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 3;
    }
}
