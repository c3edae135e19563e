// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product: only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may use anything under oracle/.
//
// Scalar CPU restatement of the reference's equality solver LexLS::internal::LexLSE
// (/root/reference/include/lexls/lexlse.h), REGULARIZATION_NONE path: factorize() :117-506,
// solve() :1015-1045, solveLeastNorm_1() :1052-1131, solveLeastNorm_2() :1138-1213, ObjectiveSensitivity() :511-602 and :611-762,
// findDescentDirection() :866-987, get_v() :1560-1582, setters :1381-1552, initialize() :1672-1693.
//
// PINNING STATUS.  The reference executes every arithmetic statement inside Eigen 3, which is not
// vendored and not installed here, so the reference cannot be compiled in this environment; its
// test-suite stores no factor / pivot / multiplier vectors.  What pins this oracle:
//   * tests/golden/test_01.dat `#Solution` (the reference's own fixture): x* of a 5-level LexLSI
//     problem, reproduced through this class + the host driver (tests/test_oracle_golden.py);
//   * the reference's closed-form multiplier KAT (interfaces/matlab-octave/tests/lexlsi/lambda_test.m);
//   * an independent numpy solver that does not follow the l-QR algorithm (oracle/oracle_np.py);
//   * the reference's manual acceptance suite of the equality solver (interfaces/matlab-octave/tests/lexlse/test_lexlse_main.m, 42 option
//     sets of test_lexlse_define.m, tol 1e-10): fixed variables, solveLeastNorm_1/2/3 and six regularization types (1, 3, 4, 5, 7, 8)
//     against the equivalent general formulation, and Tikhonov against seq_lexls.m (tests/test_oracle_golden.py::test_reference_lexlse_suite).
// Factor internals (pivot order, Householder scalars, Gauss multipliers) are "parity unpinned":
// this file DEFINES them.  Eigen primitive semantics follow the reference's in-tree MATLAB
// restatements (interfaces/matlab-octave/tests/implementation/lexqr/eigen_like_syntax/*.m).
//
// ARITHMETIC CONTRACT (what the HIP kernels reproduce bit-for-bit; compile with -ffp-contract=off):
//   dot / squaredNorm : s = 0; s = fma(a_i, b_i, s) for i ascending
//   "c -= a*b"        : c = fma(-a, b, c), inner index ascending, accumulating INTO c
//   Gauss TRSM        : L_ip = (A_ip - sum_{q<p} L_iq R_qp) * (1 / R_pp), q ascending, reciprocal computed once per pivot
//   triangular solve  : column-oriented, x_j /= R_jj then x_i = fma(-R_ij, x_j, x_i) for i < j, j descending
//   division, sqrt    : IEEE correctly rounded
//   argmax            : first occurrence of the maximum (Eigen maxCoeff, eigen_like_syntax/maxCoeff.m:11)
#pragma once

#include <cfloat>
#include <cmath>
#include <lexls/typedefs.h>

namespace lexls_oracle
{
    using namespace LexLS;

    // ---- Eigen primitives (real double), restated -------------------------------------------------

    /// squaredNorm of n strided values
    inline double sqnorm(const double *v, Index n)
    {
        double s = 0.0;
        for (Index i = 0; i < n; i++) s = std::fma(v[i], v[i], s);
        return s;
    }

    /// Eigen makeHouseholderInPlace on v[0..n) (makeHouseholderInPlace.m:32-55; Eigen >= 3.3 uses
    /// "tailSqNorm <= DBL_MIN" instead of "== 0").  On exit v[0] is untouched (caller stores beta),
    /// v[1..n) = essential part.
    inline void make_householder(double *v, Index n, double &tau, double &beta)
    {
        const double tailSq = (n <= 1) ? 0.0 : sqnorm(v + 1, n - 1);
        const double c0     = v[0];
        if (tailSq <= DBL_MIN)
        {
            tau  = 0.0;
            beta = c0;
            for (Index i = 1; i < n; i++) v[i] = 0.0;
        }
        else
        {
            beta = std::sqrt(std::fma(c0, c0, tailSq));
            if (c0 >= 0.0) beta = -beta;
            const double den = c0 - beta;
            for (Index i = 1; i < n; i++) v[i] = v[i] / den;
            tau = (beta - c0) / beta;
        }
    }

    /// Eigen applyHouseholderOnTheLeft on one column a[0..n) (applyHouseholderOnTheLeft.m:11-17)
    inline void apply_householder(const double *ess, double tau, double *a, Index n)
    {
        if (n == 1)
        {
            a[0] *= (1.0 - tau);
        }
        else if (tau != 0.0)
        {
            double tmp = 0.0;
            for (Index i = 1; i < n; i++) tmp = std::fma(ess[i - 1], a[i], tmp);
            tmp += a[0];
            a[0] = std::fma(-tau, tmp, a[0]);
            for (Index i = 1; i < n; i++) a[i] = std::fma(-(tau * ess[i - 1]), tmp, a[i]);
        }
    }

    /// Eigen JacobiRotation::makeGivens for reals (semantics: SURVEY.md section 8(a))
    inline void make_givens(double p, double q, double &c, double &s)
    {
        if (q == 0.0)
        {
            c = p < 0.0 ? -1.0 : 1.0;
            s = 0.0;
        }
        else if (p == 0.0)
        {
            c = 0.0;
            s = q < 0.0 ? 1.0 : -1.0;
        }
        else if (std::abs(p) > std::abs(q))
        {
            const double t = q / p;
            double u       = std::sqrt(std::fma(t, t, 1.0));
            if (p < 0.0) u = -u;
            c = 1.0 / u;
            s = -t * c;
        }
        else
        {
            const double t = p / q;
            double u       = std::sqrt(std::fma(t, t, 1.0));
            if (q < 0.0) u = -u;
            s = -1.0 / u;
            c = -t * s;
        }
    }

    // ---- the class --------------------------------------------------------------------------------

    class LexLSE
    {
    public:
        LexLSE() : nVar(0), nObj(0), nCtr(0), nVarFixed(0), nVarFixedInit(0), TotalRank(0) {}
        LexLSE(Index nVar_, Index nObj_, Index *ObjDim_) : nVarFixed(0), nVarFixedInit(0)
        {
            resize(nVar_, nObj_, ObjDim_);
            setObjDim(ObjDim_);
        }

        /// lexlse.h:67-103
        void resize(Index nVar_, Index nObj_, Index *maxObjDim)
        {
            nVar = nVar_;
            nObj = nObj_;
            obj_info.assign(nObj, internal::ObjectiveInfo());
            x.resize(nVar);
            Index cap = 0;
            for (Index k = 0; k < nObj; k++) cap += maxObjDim[k];
            hh_scalars.resize(cap);
            ctr_type.assign(cap, CTR_INACTIVE);
            LOD.resize(cap, nVar + 1);
            PROBLEM_DATA.resize(cap, nVar + 1);
            dWorkspace.resize(2 * std::max(cap, nVar) + nVar + 1);
            null_space.resize(nVar, nVar + 1);
            X_mu.resize(nVar, nObj);     // :96-99
            X_mu_rhs.resize(nVar, nObj);
            residual_mu.resize(cap);
            column_permutations.resize(nVar);
            nCtr      = 0;
            TotalRank = 0;
        }

        /// lexlse.h:1426-1442 (calls initialize(), :1672-1693)
        void setObjDim(Index *ObjDim_)
        {
            nCtr = 0;
            for (Index k = 0; k < nObj; k++)
            {
                nCtr += ObjDim_[k];
                obj_info[k].dim = ObjDim_[k];
                if (k > 0) obj_info[k].first_row_index = obj_info[k - 1].first_row_index + obj_info[k - 1].dim;
            }
            initialize();
        }

        void setParameters(const ParametersLexLSE &p) { parameters = p; }
        void setRegularizationFactor(Index ObjIndex, RealScalar factor) { obj_info[ObjIndex].regularization_factor = factor; }

        /// lexlse.h:1449-1462
        void setFixedVariablesCount(Index nVarFixed_)
        {
            if (nVarFixed_ > nVar) throw Exception("Cannot fix more than nVar variables");
            nVarFixed = nVarFixed_;
            fixed_var_index.resize(nVarFixed);
            fixed_var_type.assign(nVarFixed, CTR_INACTIVE);
        }

        /// lexlse.h:1381-1388
        void fixVariable(Index VarIndex, RealScalar VarValue, ConstraintActivationType type = CTR_ACTIVE_UB)
        {
            fixed_var_index(nVarFixedInit) = VarIndex;
            x(nVarFixedInit)               = VarValue;
            fixed_var_type[nVarFixedInit]  = type;
            nVarFixedInit++;
        }

        /// lexlse.h:1398-1419
        void fixVariables(Index nVarFixed_, Index *VarIndex, RealScalar *VarValue, ConstraintActivationType *type)
        {
            setFixedVariablesCount(nVarFixed_);
            for (Index k = 0; k < nVarFixed; k++)
            {
                fixed_var_index(k) = VarIndex[k];
                x(k)               = VarValue[k];
                fixed_var_type[k]  = type[k];
            }
        }

        /// lexlse.h:1511-1514 (data: capacity-rows x (nVar+1), or nCtr rows)
        void setProblem(const dMatrixConstRef &data)
        {
            for (Index j = 0; j < data.cols(); j++)
                for (Index i = 0; i < data.rows(); i++) LOD(i, j) = data(i, j);
        }

        /// lexlse.h:1522-1530
        void setData(Index ObjIndex, const dMatrixConstRef &data)
        {
            if (ObjIndex >= nObj) throw Exception("ObjIndex >= nObj");
            const Index F = obj_info[ObjIndex].first_row_index;
            for (Index j = 0; j <= nVar; j++)
                for (Index i = 0; i < obj_info[ObjIndex].dim; i++) LOD(F + i, j) = data(i, j);
        }

        /// lexlse.h:1539-1543 (row given as nVar values with a stride, because objective data is column-major)
        void setCtrStrided(Index CtrIndex, const RealScalar *row, Index stride, RealScalar rhs)
        {
            for (Index j = 0; j < nVar; j++) LOD(CtrIndex, j) = row[static_cast<size_t>(j) * stride];
            LOD(CtrIndex, nVar) = rhs;
        }
        void setCtr(Index CtrIndex, const RealScalar *row, RealScalar rhs) { setCtrStrided(CtrIndex, row, 1, rhs); }

        /// lexlse.h:1548-1552
        void setCtrType(Index ObjIndex, Index CtrIndex, ConstraintActivationType type) { ctr_type[obj_info[ObjIndex].first_row_index + CtrIndex] = type; }

        // ------------------------------------------------------------------------------------------
        /// lexlse.h:117-506
        void factorize()
        {
            const bool reg = parameters.regularization_type != REGULARIZATION_NONE;
            if (reg) check_regularization_type();

            PROBLEM_DATA = LOD; // :119
            const Index M  = nCtr;
            const Index n  = nVar;

            // fixed variables: bring their columns to the front, move their contribution to the RHS (:132-156)
            if (nVarFixed > 0)
            {
                for (Index k = 0; k < nVarFixed; k++)
                {
                    const Index coeff      = fixed_var_index(k);
                    column_permutations(k) = coeff;
                    if (k != coeff) swap_columns(k, coeff, M);
                    for (Index i = k + 1; i < nVarFixed; i++)
                    {
                        if (fixed_var_index(i) == k)
                        {
                            fixed_var_index(i) = coeff;
                            break;
                        }
                    }
                }
                for (Index i = 0; i < M; i++)
                {
                    double s = 0.0;
                    for (Index k = 0; k < nVarFixed; k++) s = std::fma(LOD(i, k), x(k), s);
                    LOD(i, n) -= s;
                }
            }

            Index ColIndex         = nVarFixed;
            Index RemainingColumns = n - nVarFixed;
            if (ColIndex >= n) // :164-175
            {
                TotalRank = nVarFixed;
                return;
            }

            double *ColNorms = dWorkspace.data(); // :178

            for (Index ObjIndex = 0; ObjIndex < nObj; ObjIndex++) // :182
            {
                const Index F   = obj_info[ObjIndex].first_row_index;
                const Index Fc  = obj_info[ObjIndex].first_col_index = ColIndex;
                const Index dim = obj_info[ObjIndex].dim;

                for (Index i = 0; i < dim; i++) residual_mu(F + i) = LOD(F + i, n); // :191 (after the eliminations, before the reflectors)

                for (Index k = ColIndex; k < n; k++) ColNorms[k] = sqnorm(&LOD(F, k), dim); // :193-196

                for (Index counter = 0; counter < dim; counter++) // :199
                {
                    const Index row = F + counter;
                    const Index R   = dim - counter;

                    // pivot: first maximum of the (down-dated) norms, then a fresh norm (:205-211)
                    Index piv = ColIndex;
                    for (Index k = ColIndex + 1; k < n; k++)
                        if (ColNorms[k] > ColNorms[piv]) piv = k;
                    const double fresh = sqnorm(&LOD(row, piv), R);
                    ColNorms[piv]      = fresh;

                    if (fresh < parameters.tol_linear_dependence) break; // :214 (squared norm vs tolerance)

                    column_permutations(ColIndex) = piv; // :222-232 (swap spans ALL nCtr rows)
                    if (ColIndex != piv)
                    {
                        swap_columns(ColIndex, piv, M);
                        std::swap(ColNorms[ColIndex], ColNorms[piv]);
                        if (reg) // :229-231 (only the rows above this level's first column)
                            for (Index i = 0; i < Fc; i++) std::swap(null_space(i, ColIndex), null_space(i, piv));
                    }

                    if (R > 1) // :239-248 (the RHS column is transformed too)
                    {
                        double tau, beta;
                        make_householder(&LOD(row, ColIndex), R, tau, beta);
                        LOD(row, ColIndex) = beta;
                        const double *ess  = &LOD(row + 1, ColIndex);
                        for (Index j = ColIndex + 1; j <= n; j++) apply_householder(ess, tau, &LOD(row, j), R);
                        hh_scalars(row) = tau;
                    }

                    ColIndex++;
                    RemainingColumns = n - ColIndex;
                    if (RemainingColumns == 0) break; // :255

                    for (Index k = ColIndex; k < n; k++) ColNorms[k] = std::fma(-LOD(row, k), LOD(row, k), ColNorms[k]); // :262-266
                }

                const Index rank = obj_info[ObjIndex].rank = ColIndex - Fc; // :272

                if (reg) regularize(ObjIndex, F, Fc, rank, RemainingColumns); // :277-411

                // Gauss step (:431-471): L <- L R^-1, Trailing -= L * Up
                if (ObjIndex < nObj - 1 && rank > 0)
                {
                    const Index Fn = F + dim;
                    // Eigen's right-side triangular solve scales by the RECIPROCAL of the diagonal
                    // (TriangularSolverMatrix.h, OnTheRight kernel: "inv_rjj = 1/rhs(j,j); r[i] *= inv_rjj")
                    double *inv_diag = dWorkspace.data() + nVar;
                    for (Index p = 0; p < rank; p++) inv_diag[p] = 1.0 / LOD(F + p, Fc + p);
                    for (Index i = Fn; i < M; i++)
                    {
                        for (Index p = 0; p < rank; p++)
                        {
                            double s = LOD(i, Fc + p);
                            for (Index q = 0; q < p; q++) s = std::fma(-LOD(i, Fc + q), LOD(F + q, Fc + p), s);
                            LOD(i, Fc + p) = s * inv_diag[p];
                        }
                        for (Index j = ColIndex; j <= n; j++)
                        {
                            double t = LOD(i, j);
                            for (Index p = 0; p < rank; p++) t = std::fma(-LOD(i, Fc + p), LOD(F + p, j), t);
                            LOD(i, j) = t;
                        }
                    }
                }

                if (RemainingColumns == 0) // :475-490
                {
                    for (Index k = ObjIndex + 1; k < nObj; k++)
                    {
                        obj_info[k].first_col_index = obj_info[k - 1].first_col_index + obj_info[k - 1].rank;
                        for (Index i = 0; i < n; i++) // :483-486
                        {
                            X_mu(i, k)     = X_mu(i, k - 1);
                            X_mu_rhs(i, k) = X_mu_rhs(i, k - 1);
                        }
                        for (Index i = 0; i < obj_info[k].dim; i++) residual_mu(obj_info[k].first_row_index + i) = -LOD(obj_info[k].first_row_index + i, n);
                    }
                    break;
                }
            }

            TotalRank = nVarFixed; // :494-498
            for (Index k = 0; k < nObj; k++) TotalRank += obj_info[k].rank;
        }

        /// lexlse.h:1015-1045
        void solve()
        {
            Index acc = 0;
            for (Index k = nObj; k--;)
            {
                const Index rank = obj_info[k].rank;
                if (rank == 0) continue;
                const Index F  = obj_info[k].first_row_index;
                const Index Fc = obj_info[k].first_col_index;
                for (Index i = 0; i < rank; i++)
                {
                    double s = LOD(F + i, nVar);
                    if (acc > 0)
                    {
                        const Index c0 = obj_info[k + 1].first_col_index;
                        for (Index j = 0; j < acc; j++) s = std::fma(-LOD(F + i, c0 + j), x(c0 + j), s);
                    }
                    x(Fc + i) = s;
                }
                back_substitute(F, Fc, rank, &x(Fc));
                acc += rank;
            }
            apply_permutation();
        }

        /// lexlse.h:1052-1131 (least-norm solution through a Givens sweep)
        void solveLeastNorm_1()
        {
            Index nVarRank = 0;
            for (Index k = 0; k < nObj; k++) nVarRank += obj_info[k].rank;
            const Index nVarFree = nVar - (nVarRank + nVarFixed);
            const Index ncol     = nVarRank + nVarFree;

            dMatrixType RT(nVarRank, ncol);
            std::vector<double> rhs(ncol, 0.0);

            Index counter = 0, col_dim = ncol;
            for (Index k = 0; k < nObj; k++) // :1081-1094
            {
                const Index F = obj_info[k].first_row_index, Fc = obj_info[k].first_col_index, rank = obj_info[k].rank;
                for (Index i = 0; i < rank; i++)
                    for (Index j = i; j < col_dim; j++) RT(counter + i, counter + j) = LOD(F + i, Fc + j);
                for (Index i = 0; i < rank; i++) rhs[counter + i] = LOD(F + i, nVar);
                counter += rank;
                col_dim -= rank;
            }

            struct Rot
            {
                double c, s;
                Index i, j;
            };
            std::vector<Rot> gs;
            gs.reserve(static_cast<size_t>(nVarFree) * nVarRank);
            for (Index i = 0; i < nVarFree; i++) // :1099-1110
            {
                for (Index j = nVarRank; j--;)
                {
                    Rot g;
                    g.i = j;
                    g.j = nVarRank + i;
                    make_givens(RT(j, j), RT(j, nVarRank + i), g.c, g.s);
                    for (Index r = 0; r <= j; r++) // applyOnTheRight on RT.topRows(j+1)
                    {
                        const double a = RT(r, g.i), b = RT(r, g.j);
                        RT(r, g.i) = std::fma(g.c, a, -(g.s * b));
                        RT(r, g.j) = std::fma(g.c, b, g.s * a);
                    }
                    gs.push_back(g);
                }
            }

            for (Index j = nVarRank; j--;) // :1115 R^-1 rhs, column-oriented like back_substitute()
            {
                rhs[j] = rhs[j] / RT(j, j);
                for (Index i = 0; i < j; i++) rhs[i] = std::fma(-RT(i, j), rhs[j], rhs[i]);
            }

            for (size_t k = gs.size(); k--;) // :1121-1124 applyOnTheLeft in reverse
            {
                const double a = rhs[gs[k].i], b = rhs[gs[k].j];
                rhs[gs[k].i] = std::fma(gs[k].c, a, gs[k].s * b);
                rhs[gs[k].j] = std::fma(gs[k].c, b, -(gs[k].s * a));
            }

            for (Index i = 0; i < ncol; i++) x(nVarFixed + i) = rhs[i]; // :1129
            apply_permutation();
        }

        /// lexlse.h:1138-1213 (least-norm solution through the normal equations of the free variables):
        /// T <- R^-1 [T | rhs];  (I + T^T T) y = T^T t_rhs by Cholesky;  x_free = y;  x_rank = R^-1 (rhs - T_LOD x_free).
        /// Arithmetic contract of the pieces Eigen leaves unspecified: products T^T T, T^T t and T_LOD x_free are ascending fma chains
        /// from 0; Cholesky is the left-looking column form (s = D_jj - sum_k L_jk^2 accumulated into s, L_ij = (D_ij - sum_k L_ik L_jk) / L_jj);
        /// both triangular solves with L are column-oriented with a true division, like back_substitute().
        void solveLeastNorm_2()
        {
            Index nVarRank = 0;
            for (Index k = 0; k < nObj; k++) nVarRank += obj_info[k].rank;
            const Index nVarFree = nVar - (nVarRank + nVarFixed);
            const Index ncol     = nVarRank + nVarFree;

            dMatrixType RT(nVarRank, ncol + 1); // [R T | rhs]
            Index counter = 0, col_dim = ncol;
            for (Index k = 0; k < nObj; k++) // :1166-1177
            {
                const Index F = obj_info[k].first_row_index, Fc = obj_info[k].first_col_index, rank = obj_info[k].rank;
                for (Index i = 0; i < rank; i++)
                {
                    for (Index j = i; j < col_dim; j++) RT(counter + i, counter + j) = LOD(F + i, Fc + j);
                    RT(counter + i, ncol) = LOD(F + i, nVar);
                }
                counter += rank;
                col_dim -= rank;
            }
            for (Index c = nVarRank; c <= ncol; c++) // :1180  T <- R^-1 T (the rhs column included), column by column
                for (Index j = nVarRank; j--;)
                {
                    RT(j, c) = RT(j, c) / RT(j, j);
                    for (Index i = 0; i < j; i++) RT(i, c) = std::fma(-RT(i, j), RT(j, c), RT(i, c));
                }
            dMatrixType D(nVarFree, nVarFree);
            std::vector<double> d(nVarFree, 0.0);
            for (Index j = 0; j < nVarFree; j++) // :1182-1188
            {
                for (Index i = j; i < nVarFree; i++)
                {
                    double acc = 0.0;
                    for (Index k = 0; k < nVarRank; k++) acc = std::fma(RT(k, nVarRank + i), RT(k, nVarRank + j), acc);
                    D(i, j) = acc;
                }
                D(j, j) += 1.0;
                double acc = 0.0;
                for (Index k = 0; k < nVarRank; k++) acc = std::fma(RT(k, nVarRank + j), RT(k, ncol), acc);
                d[j] = acc;
            }
            for (Index j = 0; j < nVarFree; j++) // :1190 LLT
            {
                double sjj = D(j, j);
                for (Index k = 0; k < j; k++) sjj = std::fma(-D(j, k), D(j, k), sjj);
                D(j, j) = std::sqrt(sjj);
                for (Index i = j + 1; i < nVarFree; i++)
                {
                    double v = D(i, j);
                    for (Index k = 0; k < j; k++) v = std::fma(-D(i, k), D(j, k), v);
                    D(i, j) = v / D(j, j);
                }
            }
            for (Index j = 0; j < nVarFree; j++) // :1191 L y = d
            {
                d[j] = d[j] / D(j, j);
                for (Index i = j + 1; i < nVarFree; i++) d[i] = std::fma(-D(i, j), d[j], d[i]);
            }
            for (Index j = nVarFree; j--;) // L^T z = y
            {
                d[j] = d[j] / D(j, j);
                for (Index i = 0; i < j; i++) d[i] = std::fma(-D(j, i), d[j], d[i]);
            }
            for (Index i = 0; i < nVarFree; i++) x(nVarFixed + nVarRank + i) = d[i];

            counter = 0;
            for (Index k = 0; k < nObj; k++) // :1193-1204
            {
                const Index F = obj_info[k].first_row_index, rank = obj_info[k].rank;
                for (Index i = 0; i < rank; i++)
                {
                    double acc = 0.0;
                    for (Index c = 0; c < nVarFree; c++) acc = std::fma(LOD(F + i, nVarRank + nVarFixed + c), x(nVarFixed + nVarRank + c), acc);
                    x(nVarFixed + counter + i) = LOD(F + i, nVar) - acc;
                }
                counter += rank;
            }
            for (Index j = nVarRank; j--;) // :1205 R^-1 on x.segment(nVarFixed, nVarRank)
            {
                x(nVarFixed + j) = x(nVarFixed + j) / RT(j, j);
                for (Index i = 0; i < j; i++) x(nVarFixed + i) = std::fma(-RT(i, j), x(nVarFixed + j), x(nVarFixed + i));
            }
            apply_permutation();
        }

        /// lexlse.h:1222-1277: least-norm solution from the null-space basis the Tikhonov family accumulates (regularization_type
        /// TIKHONOV with all factors 0): null_space(0:nVarRank, nVarFixed:) = [inv(R) | -inv(R) [T | rhs]].
        void solveLeastNorm_3()
        {
            Index nVarRank = 0;
            for (Index k = 0; k < nObj; k++) nVarRank += obj_info[k].rank;
            const Index nVarFree = nVar - (nVarRank + nVarFixed);
            const Index c0       = nVarFixed + nVarRank; // first column of T inside null_space
            Dense D(nVarFree, nVarFree);
            std::vector<double> d(nVarFree, 0.0);
            for (Index j = 0; j < nVarFree; j++)
            {
                for (Index i = j; i < nVarFree; i++)
                {
                    double acc = 0.0;
                    for (Index k = 0; k < nVarRank; k++) acc = std::fma(null_space(k, c0 + i), null_space(k, c0 + j), acc);
                    D(i, j) = acc;
                }
                D(j, j) += 1.0;
                double acc = 0.0;
                for (Index k = 0; k < nVarRank; k++) acc = std::fma(null_space(k, c0 + j), null_space(k, c0 + nVarFree), acc);
                d[j] = acc;
            }
            cholesky_solve(D, d, nVarFree);
            for (Index i = 0; i < nVarFree; i++) x(c0 + i) = d[i];
            Index counter = 0;
            for (Index k = 0; k < nObj; k++)
            {
                const Index F = obj_info[k].first_row_index, rank = obj_info[k].rank;
                for (Index i = 0; i < rank; i++)
                {
                    double acc = 0.0;
                    for (Index c = 0; c < nVarFree; c++) acc = std::fma(LOD(F + i, c0 + c), x(c0 + c), acc);
                    x(nVarFixed + counter + i) = LOD(F + i, nVar) - acc;
                }
                counter += rank;
            }
            std::vector<double> out(nVarRank);
            for (Index i = 0; i < nVarRank; i++) // x_rank <- triu(iR) * x_rank
            {
                double acc = 0.0;
                for (Index j = i; j < nVarRank; j++) acc = std::fma(null_space(i, nVarFixed + j), x(nVarFixed + j), acc);
                out[i] = acc;
            }
            for (Index i = 0; i < nVarRank; i++) x(nVarFixed + i) = out[i];
            apply_permutation();
        }

        /// lexlse.h:611-762.  On exit dWorkspace.head(nVarFixed+nLambda) = [lambda_fixed; lambda].
        bool ObjectiveSensitivity(Index ObjIndex, Index &CtrIndex2Remove, int &ObjIndex2Remove, RealScalar tol_wrong_sign_lambda,
                                  RealScalar tol_correct_sign_lambda, RealScalar &maxAbsValue)
        {
            maxAbsValue = 0.0;
            bool found  = false;

            double *LambdaFixed, *Lambda, *rhs;
            Index nLambda;
            dual_begin(ObjIndex, LambdaFixed, Lambda, rhs, nLambda);

            Index F = obj_info[ObjIndex].first_row_index, Fc = obj_info[ObjIndex].first_col_index;
            Index dim = obj_info[ObjIndex].dim;

            found = findDescentDirection(static_cast<int>(F), dim, maxAbsValue, CtrIndex2Remove, Lambda, tol_wrong_sign_lambda, tol_correct_sign_lambda);
            if (found) ObjIndex2Remove = static_cast<int>(ObjIndex);

            if (ObjIndex > 0)
            {
                dual_accumulate(F, dim, Fc, Lambda, rhs);
                for (Index k = ObjIndex; k--;)
                {
                    F   = obj_info[k].first_row_index;
                    Fc  = obj_info[k].first_col_index;
                    dim = obj_info[k].dim;
                    dual_level(k, Lambda, rhs);
                    const bool b = findDescentDirection(static_cast<int>(F), dim, maxAbsValue, CtrIndex2Remove, Lambda, tol_wrong_sign_lambda, tol_correct_sign_lambda);
                    if (b) ObjIndex2Remove = static_cast<int>(k);
                    found = found || b;
                }
            }

            if (nVarFixed > 0)
            {
                dual_fixed(nLambda, LambdaFixed, Lambda);
                const bool b = findDescentDirection(-1, nVarFixed, maxAbsValue, CtrIndex2Remove, LambdaFixed, tol_wrong_sign_lambda, tol_correct_sign_lambda);
                if (b) ObjIndex2Remove = -1;
                found = found || b;
            }
            return found;
        }

        /// lexlse.h:511-602 (all wrong-sign multipliers; used with deactivate_first_wrong_sign).
        /// The reference passes (Lambda, ObjDim) instead of (LambdaFixed, nVarFixed) for the fixed
        /// variables (:599-600); kept, but the loop is clipped to nVarFixed so it cannot run out of
        /// fixed_var_type's bounds.
        void ObjectiveSensitivity(Index ObjIndex, RealScalar tol_wrong_sign_lambda, RealScalar tol_correct_sign_lambda, std::vector<ConstraintInfo> &ctr_wrong_sign)
        {
            double *LambdaFixed, *Lambda, *rhs;
            Index nLambda;
            dual_begin(ObjIndex, LambdaFixed, Lambda, rhs, nLambda);

            Index F = obj_info[ObjIndex].first_row_index, Fc = obj_info[ObjIndex].first_col_index;
            Index dim = obj_info[ObjIndex].dim;
            findDescentDirectionAll(static_cast<int>(ObjIndex), static_cast<int>(F), dim, Lambda, tol_wrong_sign_lambda, tol_correct_sign_lambda, ctr_wrong_sign);

            if (ObjIndex > 0)
            {
                dual_accumulate(F, dim, Fc, Lambda, rhs);
                for (Index k = ObjIndex; k--;)
                {
                    F   = obj_info[k].first_row_index;
                    dim = obj_info[k].dim;
                    dual_level(k, Lambda, rhs);
                    findDescentDirectionAll(static_cast<int>(k), static_cast<int>(F), dim, Lambda, tol_wrong_sign_lambda, tol_correct_sign_lambda, ctr_wrong_sign);
                }
            }
            if (nVarFixed > 0)
            {
                dual_fixed(nLambda, LambdaFixed, Lambda);
                findDescentDirectionAll(-1, -1, std::min(dim, nVarFixed), Lambda, tol_wrong_sign_lambda, tol_correct_sign_lambda, ctr_wrong_sign);
            }
        }

        /// lexlse.h:1560-1582: residuals v = A x - b recovered through Q; valid in dWorkspace.head(nCtr)
        dVectorType &get_v()
        {
            double *v = dWorkspace.data();
            for (Index k = 0; k < nObj; k++)
            {
                const Index F = obj_info[k].first_row_index, dim = obj_info[k].dim, rank = obj_info[k].rank;
                for (Index i = 0; i < rank; i++) v[F + i] = 0.0;
                for (Index i = rank; i < dim; i++) v[F + i] = -LOD(F + i, nVar);
                apply_q(k, v + F);
            }
            return dWorkspace;
        }

        const dVectorType &get_x() const { return x; }
        Index getDim(Index k) const { return obj_info[k].dim; }
        Index getRank(Index k) const { return obj_info[k].rank; }
        Index getFirstColIndex(Index k) const { return obj_info[k].first_col_index; }
        Index getFirstRowIndex(Index k) const { return obj_info[k].first_row_index; }
        Index get_nObj() const { return nObj; }
        Index get_nVar() const { return nVar; }
        Index get_nCtr() const { return nCtr; }
        Index getTotalRank() const { return TotalRank; }
        Index getFixedVariablesCount() const { return nVarFixed; }
        const iVectorType &getFixedVarIndex() const { return fixed_var_index; }
        const dVectorType &getWorkspace() const { return dWorkspace; }
        const dMatrixType &get_lexqr() const { return LOD; }
        const dMatrixType &get_X_mu() const { return X_mu; }         // :1636-1650
        const dMatrixType &get_X_mu_rhs() const { return X_mu_rhs; }
        const dVectorType &get_residual_mu() const { return residual_mu; }
        const dMatrixType &get_data() const { return PROBLEM_DATA; }
        const dVectorType &get_hh_scalars() const { return hh_scalars; }
        const iVectorType &get_column_permutations() const { return column_permutations; }
        const std::vector<ConstraintActivationType> &get_ctr_type() const { return ctr_type; }

        /// lexlse.h:1654-1658
        void reset()
        {
            initialize();
            for (Index k = 0; k < nVarFixed; k++) x(k) = 0.0;
        }

    private:
        /// lexlse.h:1672-1693
        void initialize()
        {
            nVarFixedInit = 0;
            TotalRank     = 0;
            for (Index k = 0; k < nObj; k++)
            {
                obj_info[k].rank            = 0;
                obj_info[k].first_col_index = 0;
            }
            hh_scalars.setZero();
            null_space.setZero(); // :1686
            X_mu.setZero();       // :1687-1689
            X_mu_rhs.setZero();
            residual_mu.setZero();
            for (Index i = nVarFixed; i < nVar; i++) x(i) = 0.0;
        }

        // ------------------------------------------------------------------------------------------
        // Regularization family (lexlse.h:277-411, :1700-2554, :2592-2625).  Restated: TIKHONOV (1), TIKHONOV_CG (2), R (3), R_NO_Z (4),
        // RT_NO_Z (5), RT_NO_Z_CG (6), TIKHONOV_1 (7, the reference's experimental type: regularize_tikhonov_1_test with the X_mu /
        // residual_mu by-products that its ObjectiveSensitivity then uses, :647-651, :799-803), TIKHONOV_2 (8), TEST (9).
        // Arithmetic contract of what Eigen leaves open: every product entry is an ascending fma chain from 0 over the contraction
        // index; "X += s * P" is fma(s, p, x) on the finished entry p; "X -= A*B" accumulates fma(-a, b, x) into x (as the Gauss
        // update does); right-side triangular solves scale by the reciprocal of the diagonal (as the Gauss TRSM does); Cholesky and
        // its two solves as in solveLeastNorm_2().
        // ------------------------------------------------------------------------------------------
        void check_regularization_type() const
        {
            switch (parameters.regularization_type)
            {
            case REGULARIZATION_TIKHONOV:
            case REGULARIZATION_TIKHONOV_CG:
            case REGULARIZATION_R:
            case REGULARIZATION_R_NO_Z:
            case REGULARIZATION_RT_NO_Z:
            case REGULARIZATION_RT_NO_Z_CG:
            case REGULARIZATION_TIKHONOV_1:
            case REGULARIZATION_TIKHONOV_2:
            case REGULARIZATION_TEST: return;
            default: throw Exception("oracle: unknown regularization type");
            }
        }

        struct Dense // small column-major scratch matrix
        {
            Index m;
            std::vector<double> a;
            Dense(Index m_, Index n_) : m(m_ ? m_ : 1), a(static_cast<size_t>(m_ ? m_ : 1) * (n_ ? n_ : 1), 0.0) {}
            double &operator()(Index i, Index j) { return a[i + static_cast<size_t>(j) * m]; }
            double operator()(Index i, Index j) const { return a[i + static_cast<size_t>(j) * m]; }
        };

        /// LLT of the lower triangle of D (N x N) in place, then D z = d in place
        static void cholesky_solve(Dense &D, std::vector<double> &d, Index N)
        {
            for (Index j = 0; j < N; j++)
            {
                double sjj = D(j, j);
                for (Index k = 0; k < j; k++) sjj = std::fma(-D(j, k), D(j, k), sjj);
                D(j, j) = std::sqrt(sjj);
                for (Index i = j + 1; i < N; i++)
                {
                    double v = D(i, j);
                    for (Index k = 0; k < j; k++) v = std::fma(-D(i, k), D(j, k), v);
                    D(i, j) = v / D(j, j);
                }
            }
            for (Index j = 0; j < N; j++)
            {
                d[j] = d[j] / D(j, j);
                for (Index i = j + 1; i < N; i++) d[i] = std::fma(-D(i, j), d[j], d[i]);
            }
            for (Index j = N; j--;)
            {
                d[j] = d[j] / D(j, j);
                for (Index i = 0; i < j; i++) d[i] = std::fma(-D(j, i), d[j], d[i]);
            }
        }

        // views of the level: R(i,j) = LOD(F+i, Fc+j) (upper part), T(i,c) = LOD(F+i, Fc+rank+c), rhs_i = LOD(F+i, nVar)
        /// lower triangle of R^T R (only the upper triangle of the block is read, lexlse.h:2156)
        void lower_RtR(Dense &D, Index F, Index Fc, Index rank) const
        {
            for (Index j = 0; j < rank; j++)
                for (Index i = j; i < rank; i++)
                {
                    double acc = 0.0;
                    for (Index k = 0; k <= j; k++) acc = std::fma(LOD(F + k, Fc + i), LOD(F + k, Fc + j), acc);
                    D(i, j) = acc;
                }
        }
        /// lower triangle of R R^T + T T^T (lexlse.h:2211-2212, :2098-2100)
        void lower_RRt_TTt(Dense &D, Index F, Index Fc, Index rank, Index RC) const
        {
            for (Index j = 0; j < rank; j++)
                for (Index i = j; i < rank; i++)
                {
                    double acc = 0.0;
                    for (Index k = i; k < rank; k++) acc = std::fma(LOD(F + i, Fc + k), LOD(F + j, Fc + k), acc);
                    double t = 0.0;
                    for (Index c = 0; c < RC; c++) t = std::fma(LOD(F + i, Fc + rank + c), LOD(F + j, Fc + rank + c), t);
                    D(i, j) = acc + t;
                }
        }
        /// out_i = sum_{k <= i} R(k,i) rhs_k   (R^T rhs)
        double Rt_rhs(Index F, Index Fc, Index i) const
        {
            double acc = 0.0;
            for (Index k = 0; k <= i; k++) acc = std::fma(LOD(F + k, Fc + i), LOD(F + k, nVar), acc);
            return acc;
        }
        /// out_i = sum_{j >= i} R(i,j) d_j      (R d)
        double R_times(Index F, Index Fc, Index rank, Index i, const std::vector<double> &d) const
        {
            double acc = 0.0;
            for (Index j = i; j < rank; j++) acc = std::fma(LOD(F + i, Fc + j), d[j], acc);
            return acc;
        }
        /// d <- sym(D) d with only the lower triangle of D stored
        static void symv_lower(const Dense &D, std::vector<double> &d, Index N)
        {
            std::vector<double> out(N, 0.0);
            for (Index i = 0; i < N; i++)
            {
                double acc = 0.0;
                for (Index j = 0; j < N; j++) acc = std::fma(i >= j ? D(i, j) : D(j, i), d[j], acc);
                out[i] = acc;
            }
            d = out;
        }

        void regularize(Index ObjIndex, Index F, Index Fc, Index rank, Index RC)
        {
            // :277-311
            if (parameters.variable_regularization_factor == 0.0)
                aRegularizationFactor = obj_info[ObjIndex].regularization_factor;
            else
            {
                aRegularizationFactor = 0.0;
                if (rank > 0)
                {
                    std::vector<double> t(rank);
                    for (Index i = 0; i < rank; i++) t[i] = LOD(F + i, nVar);
                    double ce = sqnorm(t.data(), rank);
                    back_substitute(F, Fc, rank, t.data());
                    ce /= sqnorm(t.data(), rank);
                    const double eps = parameters.variable_regularization_factor;
                    if (ce < eps)
                    {
                        aRegularizationFactor = std::sqrt(1 - (ce * ce) / (eps * eps));
                        aRegularizationFactor *= obj_info[ObjIndex].regularization_factor;
                    }
                }
            }
            const bool nonzero = !(std::abs(aRegularizationFactor - 0.0) < 1e-15); // utility.h:48-51
            switch (parameters.regularization_type) // :314-395
            {
            case REGULARIZATION_TIKHONOV:
                if (nonzero)
                {
                    if (Fc + rank <= RC)
                        regularize_tikhonov_2(F, Fc, rank, RC);
                    else
                        regularize_tikhonov_1(F, Fc, rank, RC);
                }
                accumulate_nullspace_basis(F, Fc, rank, RC);
                break;
            case REGULARIZATION_TIKHONOV_1: // :378-387
                if (nonzero) regularize_tikhonov_1_test(F, Fc, rank, RC, ObjIndex);
                accumulate_nullspace_basis(F, Fc, rank, RC);
                break;
            case REGULARIZATION_TIKHONOV_2:
                if (nonzero) regularize_tikhonov_2(F, Fc, rank, RC);
                accumulate_nullspace_basis(F, Fc, rank, RC);
                break;
            case REGULARIZATION_TIKHONOV_CG: // :337-345
                if (nonzero) regularize_cg(F, Fc, rank, RC, true);
                accumulate_nullspace_basis(F, Fc, rank, RC);
                break;
            case REGULARIZATION_RT_NO_Z_CG: // :371-377
                if (nonzero) regularize_cg(F, Fc, rank, RC, false);
                break;
            case REGULARIZATION_R:
                if (nonzero) regularize_R(F, Fc, rank);
                accumulate_nullspace_basis(F, Fc, rank, RC);
                break;
            case REGULARIZATION_R_NO_Z:
                if (nonzero) regularize_R_NO_Z(F, Fc, rank);
                break;
            case REGULARIZATION_RT_NO_Z:
                if (nonzero) regularize_RT_NO_Z(F, Fc, rank, RC);
                break;
            case REGULARIZATION_TEST:
                if (nonzero)
                    for (Index i = 0; i < rank; i++) LOD(F + i, nVar) *= aRegularizationFactor; // :2244
                break;
            default: break;
            }
        }

        /// lexlse.h:1700-1760
        void regularize_tikhonov_1(Index F, Index Fc, Index rank, Index RC, std::vector<double> *solution = NULL)
        {
            const double mu = aRegularizationFactor * aRegularizationFactor;
            const Index m0 = Fc - nVarFixed, N = RC + rank;
            Dense D(N, N);
            std::vector<double> d(N, 0.0);
            lower_RtR(D, F, Fc, rank);
            for (Index b = 0; b < RC; b++) // Tk'*Tk (lower)
                for (Index a = b; a < RC; a++)
                {
                    double acc = 0.0;
                    for (Index k = 0; k < rank; k++) acc = std::fma(LOD(F + k, Fc + rank + a), LOD(F + k, Fc + rank + b), acc);
                    D(rank + a, rank + b) = acc;
                }
            for (Index j = 0; j < rank; j++) // Tk'*triu(Rk)
                for (Index a = 0; a < RC; a++)
                {
                    double acc = 0.0;
                    for (Index k = 0; k <= j; k++) acc = std::fma(LOD(F + k, Fc + rank + a), LOD(F + k, Fc + j), acc);
                    D(rank + a, j) = acc;
                }
            for (Index j = 0; j < N; j++) // += mu * up'*up, up = null_space(0:m0, Fc:Fc+N)
                for (Index i = j; i < N; i++)
                {
                    double acc = 0.0;
                    for (Index r = 0; r < m0; r++) acc = std::fma(null_space(r, Fc + i), null_space(r, Fc + j), acc);
                    D(i, j) = std::fma(mu, acc, D(i, j));
                }
            for (Index i = 0; i < N; i++) D(i, i) += mu;
            for (Index i = 0; i < rank; i++) d[i] = Rt_rhs(F, Fc, i);
            for (Index a = 0; a < RC; a++)
            {
                double acc = 0.0;
                for (Index k = 0; k < rank; k++) acc = std::fma(LOD(F + k, Fc + rank + a), LOD(F + k, nVar), acc);
                d[rank + a] = acc;
            }
            for (Index i = 0; i < N; i++)
            {
                double acc = 0.0;
                for (Index r = 0; r < m0; r++) acc = std::fma(null_space(r, Fc + i), null_space(r, nVar), acc);
                d[i] = std::fma(mu, acc, d[i]);
            }
            cholesky_solve(D, d, N);
            std::vector<double> out(rank);
            for (Index i = 0; i < rank; i++)
            {
                double t = 0.0;
                for (Index c = 0; c < RC; c++) t = std::fma(LOD(F + i, Fc + rank + c), d[rank + c], t);
                out[i] = R_times(F, Fc, rank, i, d) + t;
            }
            for (Index i = 0; i < rank; i++) LOD(F + i, nVar) = out[i];
            if (solution) solution->swap(d);
        }

        /// lexlse.h:1774-1886: regularize_tikhonov_1 plus the by-products of the experimental type 7 — the residual of the regularized
        /// level (residual_mu), the regularized solution of the levels 0..ObjIndex (a column of X_mu, get_intermediate_x :2010-2071).
        /// The permutation at the end counts the ranks without the fixed variables, as the reference does ("WARNING: how about fixed
        /// variables", :189).
        void regularize_tikhonov_1_test(Index F, Index Fc, Index rank, Index RC, Index ObjIndex)
        {
            const Index dim = obj_info[ObjIndex].dim, N = RC + rank;
            std::vector<double> d;
            regularize_tikhonov_1(F, Fc, rank, RC, &d);

            std::vector<double> w(dim, 0.0); // Q1 [R T] d - b  (:1848-1854)
            for (Index i = 0; i < rank; i++) w[i] = LOD(F + i, nVar);
            apply_q(ObjIndex, w.data());
            for (Index i = 0; i < dim; i++) residual_mu(F + i) = w[i] - residual_mu(F + i);

            for (Index i = 0; i < N; i++) X_mu(nVar - N + i, ObjIndex) = d[i]; // :1857
            for (Index i = 0; i < ObjIndex; i++)                               // get_intermediate_x, :2026-2040
            {
                const Index Fi = obj_info[i].first_row_index, Fci = obj_info[i].first_col_index, ri = obj_info[i].rank;
                for (Index r = 0; r < ri; r++)
                {
                    double acc = 0.0;
                    for (Index c = 0; c < N; c++) acc = std::fma(LOD(Fi + r, nVar - N + c), X_mu(nVar - N + c, ObjIndex), acc);
                    X_mu(Fci + r, ObjIndex) = LOD(Fi + r, nVar) - acc;
                }
            }
            Index acc_ranks = 0;
            for (Index k = ObjIndex; k--;) // :2046-2070
            {
                const Index Fk = obj_info[k].first_row_index, Fck = obj_info[k].first_col_index, rk = obj_info[k].rank;
                if (rk == 0) continue;
                if (acc_ranks > 0)
                {
                    const Index c0 = obj_info[k + 1].first_col_index;
                    for (Index i = 0; i < rk; i++)
                    {
                        double s2 = X_mu(Fck + i, ObjIndex);
                        for (Index j = 0; j < acc_ranks; j++) s2 = std::fma(-LOD(Fk + i, c0 + j), X_mu(c0 + j, ObjIndex), s2);
                        X_mu(Fck + i, ObjIndex) = s2;
                    }
                }
                back_substitute(Fk, Fck, rk, &X_mu(Fck, ObjIndex));
                acc_ranks += rk;
            }
            Index total = 0; // :1863-1874
            for (Index k = 0; k <= ObjIndex; k++) total += obj_info[k].rank;
            for (Index k = total; k--;) std::swap(X_mu(k, ObjIndex), X_mu(column_permutations(k), ObjIndex));
        }

        /// lexlse.h:1921-1959: right-hand side of the dual solve of the experimental type 7
        void initialize_rhs(Index ObjIndex, double *rhs, Index rhs_size)
        {
            const double f = obj_info[ObjIndex].regularization_factor;
            aRegularizationFactor = f;
            double *c = &X_mu_rhs(0, ObjIndex);
            for (Index i = 0; i < nVar; i++) c[i] = X_mu(i, ObjIndex);
            for (Index k = 0; k < TotalRank; k++) std::swap(c[k], c[column_permutations(k)]); // P' * .
            for (Index i = 0; i < nVar; i++) c[i] *= -f * f;
            const Index last = obj_info[ObjIndex].first_col_index + obj_info[ObjIndex].rank; // one past the last column of interest
            for (Index k = 0; k <= ObjIndex; k++)
            {
                const Index Fk = obj_info[k].first_row_index, Fck = obj_info[k].first_col_index, rk = obj_info[k].rank;
                if (k > 0)
                {
                    const Index Fp = obj_info[k - 1].first_row_index, Fcp = obj_info[k - 1].first_col_index, rp = obj_info[k - 1].rank;
                    const Index remain = last - Fck;
                    for (Index j = 0; j < remain; j++)
                    {
                        double s2 = c[Fck + j];
                        for (Index i = 0; i < rp; i++) s2 = std::fma(-LOD(Fp + i, Fck + j), c[Fcp + i], s2);
                        c[Fck + j] = s2;
                    }
                }
                for (Index j = 0; j < rk; j++) // R_k' z = c (forward substitution with the transposed upper triangle)
                {
                    double s2 = c[Fck + j];
                    for (Index i = 0; i < j; i++) s2 = std::fma(-LOD(Fk + i, Fck + j), c[Fck + i], s2);
                    c[Fck + j] = s2 / LOD(Fk + j, Fck + j);
                }
            }
            for (Index i = 0; i < rhs_size; i++) rhs[i] = c[i];
        }

        /// lexlse.h:2076-2133
        void regularize_tikhonov_2(Index F, Index Fc, Index rank, Index RC)
        {
            const double f = aRegularizationFactor, mu = f * f;
            const Index m0 = Fc - nVarFixed, N = m0 + rank, W = RC + rank;
            Dense D(N, N);
            std::vector<double> d(N, 0.0);
            lower_RRt_TTt(D, F, Fc, rank, RC);
            for (Index t = 0; t < m0; t++) // mu * up*up' (lower)
                for (Index s2 = t; s2 < m0; s2++)
                {
                    double acc = 0.0;
                    for (Index c = 0; c < W; c++) acc = std::fma(null_space(s2, Fc + c), null_space(t, Fc + c), acc);
                    D(rank + s2, rank + t) = mu * acc;
                }
            for (Index i = 0; i < rank; i++) // f * (up.leftCols(rank)*triu(Rk)' + up.rightCols(RC)*Tk')
                for (Index s2 = 0; s2 < m0; s2++)
                {
                    double a1 = 0.0;
                    for (Index c = i; c < rank; c++) a1 = std::fma(null_space(s2, Fc + c), LOD(F + i, Fc + c), a1);
                    double a2 = 0.0;
                    for (Index c = 0; c < RC; c++) a2 = std::fma(null_space(s2, Fc + rank + c), LOD(F + i, Fc + rank + c), a2);
                    D(rank + s2, i) = std::fma(f, a2, f * a1);
                }
            for (Index i = 0; i < N; i++) D(i, i) += mu;
            for (Index i = 0; i < rank; i++) d[i] = LOD(F + i, nVar);
            for (Index s2 = 0; s2 < m0; s2++) d[rank + s2] = f * null_space(s2, nVar);
            Dense D0 = D; // Eigen::LLT works on a copy (:2121); the matrix itself is needed again below
            cholesky_solve(D, d, N);
            for (Index i = 0; i < N; i++) D0(i, i) -= mu;
            symv_lower(D0, d, N);
            for (Index i = 0; i < rank; i++) LOD(F + i, nVar) = d[i];
        }

        /// lexlse.h:2138-2170
        void regularize_R(Index F, Index Fc, Index rank)
        {
            const double mu = aRegularizationFactor * aRegularizationFactor;
            const Index m0 = Fc - nVarFixed;
            Dense D(rank, rank);
            std::vector<double> d(rank, 0.0);
            lower_RtR(D, F, Fc, rank);
            for (Index j = 0; j < rank; j++)
                for (Index i = j; i < rank; i++)
                {
                    double acc = 0.0;
                    for (Index r = 0; r < m0; r++) acc = std::fma(null_space(r, Fc + i), null_space(r, Fc + j), acc);
                    D(i, j) = std::fma(mu, acc, D(i, j));
                }
            for (Index i = 0; i < rank; i++) D(i, i) += mu;
            for (Index i = 0; i < rank; i++)
            {
                double acc = 0.0;
                for (Index r = 0; r < m0; r++) acc = std::fma(null_space(r, Fc + i), null_space(r, nVar), acc);
                d[i] = mu * acc + Rt_rhs(F, Fc, i);
            }
            cholesky_solve(D, d, rank);
            std::vector<double> out(rank);
            for (Index i = 0; i < rank; i++) out[i] = R_times(F, Fc, rank, i, d);
            for (Index i = 0; i < rank; i++) LOD(F + i, nVar) = out[i];
        }

        /// lexlse.h:2175-2200
        void regularize_R_NO_Z(Index F, Index Fc, Index rank)
        {
            const double mu = aRegularizationFactor * aRegularizationFactor;
            Dense D(rank, rank);
            std::vector<double> d(rank, 0.0);
            lower_RtR(D, F, Fc, rank);
            for (Index i = 0; i < rank; i++) D(i, i) += mu;
            for (Index i = 0; i < rank; i++) d[i] = Rt_rhs(F, Fc, i);
            cholesky_solve(D, d, rank);
            std::vector<double> out(rank);
            for (Index i = 0; i < rank; i++) out[i] = R_times(F, Fc, rank, i, d);
            for (Index i = 0; i < rank; i++) LOD(F + i, nVar) = out[i];
        }

        /// lexlse.h:2205-2236
        void regularize_RT_NO_Z(Index F, Index Fc, Index rank, Index RC)
        {
            const double mu = aRegularizationFactor * aRegularizationFactor;
            Dense D(rank, rank);
            std::vector<double> d(rank, 0.0);
            lower_RRt_TTt(D, F, Fc, rank, RC);
            for (Index i = 0; i < rank; i++) D(i, i) += mu;
            for (Index i = 0; i < rank; i++) d[i] = LOD(F + i, nVar);
            // the Cholesky factor overwrites D: keep the matrix for the product below
            Dense D0 = D;
            cholesky_solve(D, d, rank);
            for (Index i = 0; i < rank; i++) D0(i, i) -= mu;
            symv_lower(D0, d, rank);
            for (Index i = 0; i < rank; i++) LOD(F + i, nVar) = d[i];
        }

        /// regularize_tikhonov_CG (lexlse.h:2256-2279, cg_tikhonov :2370-2461) with with_z, regularize_RT_NO_Z_CG (:2325-2347, cg_RT
        /// :2472-2554) without: CGLS on  [Rk Tk; f Sk; f I] x = [y; f s; 0]  (the Sk rows only with_z), started from x = 0, at most
        /// max_number_of_CG_iterations steps, tolerance 1e-12 on ||s||; then rhs <- [Rk Tk] x.
        /// Contract: every matrix-vector entry is an ascending fma chain from 0 that is then added / subtracted as the reference's
        /// statement reads; "v += a*w" is fma(a, w, v); squared norms are ascending chains.
        void regularize_cg(Index F, Index Fc, Index rank, Index RC, bool with_z)
        {
            const double f  = aRegularizationFactor;
            const Index m0  = with_z ? Fc - nVarFixed : 0, N = rank + RC;
            std::vector<double> x(N, 0.0), r1(rank), r2(m0), r3(N), q1(rank), q2(m0), q3(N), sv(N), pv(N);
            auto RTx = [&](const std::vector<double> &v, Index i, double &t_out, double &r_out) { // (Tk v_tail)_i and (triu(Rk) v_head)_i
                double t = 0.0;
                for (Index c = 0; c < RC; c++) t = std::fma(LOD(F + i, Fc + rank + c), v[rank + c], t);
                double rr = 0.0;
                for (Index j = i; j < rank; j++) rr = std::fma(LOD(F + i, Fc + j), v[j], rr);
                t_out = t;
                r_out = rr;
            };
            auto compute_s = [&]() { // s = [Rk Tk; f Sk; f I]' r
                for (Index i = 0; i < N; i++)
                {
                    double acc = 0.0;
                    for (Index k = 0; k < m0; k++) acc = std::fma(null_space(k, Fc + i), r2[k], acc);
                    sv[i] = with_z ? (acc + r3[i]) * f : f * r3[i];
                }
                for (Index i = 0; i < rank; i++)
                {
                    double acc = 0.0;
                    for (Index k = 0; k <= i; k++) acc = std::fma(LOD(F + k, Fc + i), r1[k], acc);
                    sv[i] += acc;
                }
                for (Index c = 0; c < RC; c++)
                {
                    double acc = 0.0;
                    for (Index k = 0; k < rank; k++) acc = std::fma(LOD(F + k, Fc + rank + c), r1[k], acc);
                    sv[rank + c] += acc;
                }
            };
            for (Index i = 0; i < rank; i++) // r = [y; f s; 0] - [Rk Tk; f Sk; f I] x   (x = 0)
            {
                double t, rr;
                RTx(x, i, t, rr);
                r1[i] = LOD(F + i, nVar) - t;
                r1[i] -= rr;
            }
            for (Index k = 0; k < m0; k++)
            {
                double acc = 0.0;
                for (Index i = 0; i < N; i++) acc = std::fma(null_space(k, Fc + i), x[i], acc);
                r2[k] = (null_space(k, nVar) - acc) * f;
            }
            for (Index i = 0; i < N; i++) r3[i] = -f * x[i];
            compute_s();
            pv           = sv;
            double gamma = sqnorm(sv.data(), N);
            Index iter   = 0;
            while (std::sqrt(gamma) > 1e-12 && iter < parameters.max_number_of_CG_iterations)
            {
                for (Index i = 0; i < rank; i++)
                {
                    double t, rr;
                    RTx(pv, i, t, rr);
                    q1[i] = t;
                    q1[i] += rr;
                }
                for (Index k = 0; k < m0; k++)
                {
                    double acc = 0.0;
                    for (Index i = 0; i < N; i++) acc = std::fma(null_space(k, Fc + i), pv[i], acc);
                    q2[k] = acc * f;
                }
                for (Index i = 0; i < N; i++) q3[i] = f * pv[i];
                double qq = 0.0; // q.squaredNorm(): one chain over [q1; q2; q3]
                for (Index i = 0; i < rank; i++) qq = std::fma(q1[i], q1[i], qq);
                for (Index k = 0; k < m0; k++) qq = std::fma(q2[k], q2[k], qq);
                for (Index i = 0; i < N; i++) qq = std::fma(q3[i], q3[i], qq);
                const double alpha = gamma / qq;
                for (Index i = 0; i < N; i++) x[i] = std::fma(alpha, pv[i], x[i]);
                for (Index i = 0; i < rank; i++) r1[i] = std::fma(-alpha, q1[i], r1[i]);
                for (Index k = 0; k < m0; k++) r2[k] = std::fma(-alpha, q2[k], r2[k]);
                for (Index i = 0; i < N; i++) r3[i] = std::fma(-alpha, q3[i], r3[i]);
                compute_s();
                const double gamma_previous = gamma;
                gamma                       = sqnorm(sv.data(), N);
                const double beta           = gamma / gamma_previous;
                for (Index i = 0; i < N; i++) pv[i] = std::fma(beta, pv[i], sv[i]);
                iter++;
            }
            std::vector<double> out(rank);
            for (Index i = 0; i < rank; i++)
            {
                double t, rr;
                RTx(x, i, t, rr);
                out[i] = rr + t;
            }
            for (Index i = 0; i < rank; i++) LOD(F + i, nVar) = out[i];
        }

        /// lexlse.h:2592-2625
        void accumulate_nullspace_basis(Index F, Index Fc, Index rank, Index RC)
        {
            const Index m0 = Fc - nVarFixed, rows = m0 + rank;
            for (Index i = 0; i < rank; i++) // LeftBlock.block(m0, 0, rank, rank).setIdentity()
                for (Index j = 0; j < rank; j++) null_space(m0 + i, Fc + j) = (i == j) ? 1.0 : 0.0;
            if (rank == 0) return;
            std::vector<double> inv_diag(rank);
            for (Index p = 0; p < rank; p++) inv_diag[p] = 1.0 / LOD(F + p, Fc + p);
            for (Index i = 0; i < rows; i++) // LeftBlock <- LeftBlock * R^-1
                for (Index p = 0; p < rank; p++)
                {
                    double sv = null_space(i, Fc + p);
                    for (Index q = 0; q < p; q++) sv = std::fma(-null_space(i, Fc + q), LOD(F + q, Fc + p), sv);
                    null_space(i, Fc + p) = sv * inv_diag[p];
                }
            for (Index i = 0; i < rows; i++) // TrailingBlock -= LeftBlock * UpBlock (RC + 1 columns: the RHS column too)
                for (Index k = 0; k <= RC; k++)
                {
                    double t = null_space(i, Fc + rank + k);
                    for (Index p = 0; p < rank; p++) t = std::fma(-null_space(i, Fc + p), LOD(F + p, Fc + rank + k), t);
                    null_space(i, Fc + rank + k) = t;
                }
        }

        void swap_columns(Index a, Index b, Index rows)
        {
            for (Index i = 0; i < rows; i++) std::swap(LOD(i, a), LOD(i, b));
        }

        /// x[0..rank) <- R^-1 x, R = LOD(F.., Fc..) upper triangular
        void back_substitute(Index F, Index Fc, Index rank, double *xk) const
        {
            // column-oriented (as Eigen's column-major triangular solver): x_j is final once every
            // column to its right has been eliminated; contributions reach x_i in order j = rank-1 .. i+1
            for (Index j = rank; j--;)
            {
                xk[j] = xk[j] / LOD(F + j, Fc + j);
                for (Index i = 0; i < j; i++) xk[i] = std::fma(-LOD(F + i, Fc + j), xk[j], xk[i]);
            }
        }

        /// x = P x with P = T(0,p0) T(1,p1) ... (lexlse.h:494-504, :1044, equivalence stated at :1869-1884)
        void apply_permutation()
        {
            for (Index k = TotalRank; k--;) std::swap(x(k), x(column_permutations(k)));
        }

        /// v <- Q_k v = H_0 H_1 ... H_{r-1} v (householderSequence, applyOnTheLeft.m:11-14)
        void apply_q(Index k, double *v) const
        {
            const Index F = obj_info[k].first_row_index, Fc = obj_info[k].first_col_index, dim = obj_info[k].dim, rank = obj_info[k].rank;
            for (Index j = rank; j--;) apply_householder(LOD.data() + (F + j + 1) + static_cast<size_t>(Fc + j) * LOD.rows(), hh_scalars(F + j), v + j, dim - j);
        }

        /// common prologue of the dual solve (:625-664): zero the workspace, residual of ObjIndex
        void dual_begin(Index ObjIndex, double *&LambdaFixed, double *&Lambda, double *&rhs, Index &nLambda)
        {
            nLambda     = 0;
            Index nRank = 0;
            for (Index k = 0; k < ObjIndex; k++)
            {
                nLambda += obj_info[k].dim;
                nRank += obj_info[k].rank;
            }
            nLambda += obj_info[ObjIndex].dim;

            double *w = dWorkspace.data();
            for (Index i = 0; i < nVarFixed + nLambda + nRank + nVarFixed; i++) w[i] = 0.0;
            LambdaFixed = w;
            Lambda      = w + nVarFixed;
            rhs         = w + nVarFixed + nLambda;

            const Index F = obj_info[ObjIndex].first_row_index, dim = obj_info[ObjIndex].dim, rank = obj_info[ObjIndex].rank;
            if (parameters.regularization_type == REGULARIZATION_TIKHONOV_1) // :647-651, :688-690 ("WARNING: TESTING" in the reference)
            {
                initialize_rhs(ObjIndex, rhs, nRank + nVarFixed);
                for (Index i = 0; i < dim; i++) Lambda[F + i] = residual_mu(F + i);
                return;
            }
            for (Index i = rank; i < dim; i++) Lambda[F + i] = -LOD(F + i, nVar);
            apply_q(ObjIndex, Lambda + F);
        }

        /// rhs.head(ColDim) -= LOD(F:F+dim, 0:ColDim)^T * Lambda.segment(F, dim)   (:706-707, :727-728)
        void dual_accumulate(Index F, Index dim, Index ColDim, const double *Lambda, double *rhs) const
        {
            for (Index c = 0; c < ColDim; c++)
            {
                double s = 0.0;
                for (Index i = 0; i < dim; i++) s = std::fma(LOD(F + i, c), Lambda[F + i], s);
                rhs[c] -= s;
            }
        }

        /// one level of the back-propagation (:712-728)
        void dual_level(Index k, double *Lambda, double *rhs) const
        {
            const Index F = obj_info[k].first_row_index, Fc = obj_info[k].first_col_index, dim = obj_info[k].dim, rank = obj_info[k].rank;
            for (Index i = 0; i < rank; i++) Lambda[F + i] = rhs[Fc + i];
            apply_q(k, Lambda + F);
            dual_accumulate(F, dim, Fc, Lambda, rhs);
        }

        /// LambdaFixed = -LOD(0:nLambda, 0:nVarFixed)^T * Lambda   (:744)
        void dual_fixed(Index nLambda, double *LambdaFixed, const double *Lambda) const
        {
            for (Index c = 0; c < nVarFixed; c++)
            {
                double s = 0.0;
                for (Index i = 0; i < nLambda; i++) s = std::fma(LOD(i, c), Lambda[i], s);
                LambdaFixed[c] = -s;
            }
        }

        /// lexlse.h:935-987
        bool findDescentDirection(int FirstRowIndex, Index ObjDim, RealScalar &maxAbsValue, Index &CtrIndex, const double *lambda, RealScalar tol_wrong,
                                  RealScalar tol_correct)
        {
            bool found = false;
            for (Index k = 0; k < ObjDim; k++)
            {
                const Index ind                = (FirstRowIndex < 0) ? k : static_cast<Index>(FirstRowIndex) + k;
                ConstraintActivationType &type = (FirstRowIndex < 0) ? fixed_var_type[ind] : ctr_type[ind];
                if (type == CTR_ACTIVE_EQ || type == CORRECT_SIGN_OF_LAMBDA) continue;
                double a = lambda[ind];
                if (type == CTR_ACTIVE_LB) a = -a;
                if (a > tol_correct)
                {
                    type = CORRECT_SIGN_OF_LAMBDA;
                }
                else if (a < -tol_wrong && a < maxAbsValue)
                {
                    found       = true;
                    maxAbsValue = a;
                    CtrIndex    = k;
                }
            }
            return found;
        }

        /// lexlse.h:866-910
        void findDescentDirectionAll(int ObjIndex, int FirstRowIndex, Index ObjDim, const double *lambda, RealScalar tol_wrong, RealScalar tol_correct,
                                     std::vector<ConstraintInfo> &out)
        {
            for (Index k = 0; k < ObjDim; k++)
            {
                const Index ind                = (FirstRowIndex < 0) ? k : static_cast<Index>(FirstRowIndex) + k;
                ConstraintActivationType &type = (FirstRowIndex < 0) ? fixed_var_type[ind] : ctr_type[ind];
                if (type == CTR_ACTIVE_EQ || type == CORRECT_SIGN_OF_LAMBDA) continue;
                double a = lambda[ind];
                if (type == CTR_ACTIVE_LB) a = -a;
                if (a > tol_correct)
                    type = CORRECT_SIGN_OF_LAMBDA;
                else if (a < -tol_wrong)
                    out.push_back(ConstraintInfo(ObjIndex, static_cast<int>(k)));
            }
        }

        Index nVar, nObj, nCtr, nVarFixed, nVarFixedInit, TotalRank;
        ParametersLexLSE parameters;
        std::vector<internal::ObjectiveInfo> obj_info;
        dMatrixType X_mu, X_mu_rhs; // by-products of the experimental REGULARIZATION_TIKHONOV_1 (lexlse.h:96-99)
        dVectorType residual_mu;
        dMatrixType LOD, PROBLEM_DATA, null_space; // null_space: nVar x (nVar+1), basis accumulated for the regularization (lexlse.h:93)
        double aRegularizationFactor = 0.0;
        dVectorType x, hh_scalars, dWorkspace;
        iVectorType column_permutations, fixed_var_index;
        std::vector<ConstraintActivationType> ctr_type, fixed_var_type;
    };
} // namespace lexls_oracle
