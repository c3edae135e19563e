/*
 * Derived work: this header restates, for a different equality-solver back end, host-side interface and control flow of
 * jrl-umi3218/lexls (include/lexls/lexlse.h), whose notice is retained as its BSD 3-clause licence requires:
 *
 * Copyright 2013-2021 INRIA
 *
 * Redistribution and use in source and binary forms, with or without modification, are permitted provided that the following
 * conditions are met:
 * 1. Redistributions of source code must retain the above copyright notice, this list of conditions and the following disclaimer.
 * 2. Redistributions in binary form must reproduce the above copyright notice, this list of conditions and the following disclaimer
 *    in the documentation and/or other materials provided with the distribution.
 * 3. Neither the name of the copyright holder nor the names of its contributors may be used to endorse or promote products derived
 *    from this software without specific prior written permission.
 *
 * THIS SOFTWARE IS PROVIDED BY THE COPYRIGHT HOLDERS AND CONTRIBUTORS "AS IS" AND ANY EXPRESS OR IMPLIED WARRANTIES, INCLUDING, BUT
 * NOT LIMITED TO, THE IMPLIED WARRANTIES OF MERCHANTABILITY AND FITNESS FOR A PARTICULAR PURPOSE ARE DISCLAIMED. IN NO EVENT SHALL
 * THE COPYRIGHT HOLDER OR CONTRIBUTORS BE LIABLE FOR ANY DIRECT, INDIRECT, INCIDENTAL, SPECIAL, EXEMPLARY, OR CONSEQUENTIAL DAMAGES
 * (INCLUDING, BUT NOT LIMITED TO, PROCUREMENT OF SUBSTITUTE GOODS OR SERVICES; LOSS OF USE, DATA, OR PROFITS; OR BUSINESS
 * INTERRUPTION) HOWEVER CAUSED AND ON ANY THEORY OF LIABILITY, WHETHER IN CONTRACT, STRICT LIABILITY, OR TORT (INCLUDING NEGLIGENCE
 * OR OTHERWISE) ARISING IN ANY WAY OUT OF THE USE OF THIS SOFTWARE, EVEN IF ADVISED OF THE POSSIBILITY OF SUCH DAMAGE.
 */
// LexLS::internal::LexLSE — drop-in for the reference's equality solver, backed by liblexls_hip.
//
// Same public member functions, argument meaning and error behaviour as the reference class
// (/root/reference/include/lexls/lexlse.h:33-2886; signatures listed in SURVEY.md section 8(b)),
// but every arithmetic member function is a call through the C ABI (include/lexls_hip.h) into the
// hand-written HIP kernels; the object keeps host copies of the problem (inputs are copied, like the
// reference's setData/setCtr) and of the last results.  One object = a batch of ONE problem; the
// batched drivers use the C ABI directly.  Link with -llexls_hip.  There is no CPU fallback: without a
// GPU the constructor throws LexLS::Exception.
//
// Differences a caller can observe: Eigen types are replaced by the containers of typedefs.h
// (dMatrixConstRef is a (ptr, rows, cols, ld) view); setCtr takes a pointer to nVar doubles;
// solveGeneralNorm throws (no caller in the reference).
#pragma once

#include <lexls/typedefs.h>
#include <lexls_hip.h>

#include <cstring>
#include <limits>

namespace LexLS
{
    namespace internal
    {
        class LexLSE
        {
        public:
            LexLSE() : h(NULL), nVar(0), nObj(0), nCtr(0), cap(0), nVarFixed(0), nVarFixedInit(0), TotalRank(0), device(0) {}
            LexLSE(Index nVar_, Index nObj_, Index *ObjDim_) : h(NULL), nVarFixed(0), nVarFixedInit(0), device(0)
            {
                resize(nVar_, nObj_, ObjDim_);
                setObjDim(ObjDim_);
            }
            ~LexLSE()
            {
                if (h) lexls_lse_destroy(h);
            }
            LexLSE(const LexLSE &)            = delete;
            LexLSE &operator=(const LexLSE &) = delete;

            /// which GPU later resize() calls allocate on (default 0)
            void setDevice(int device_) { device = device_; }
            /// ObjectiveSensitivity(level) goes on through the following levels until one reports a wrong-sign multiplier (one device call
            /// for the removal search of lexlsi.h:1121-1132); the caller then treats "not found" as final (LexLSI_T::setSensitivityScansAllLevels)
            void setSensitivityScan(bool on)
            {
                sens_scan = on;
                if (h) check(lexls_lse_set_sensitivity_scan(h, on ? 1 : 0));
            }

            /// lexlse.h:67-103
            void resize(Index nVar_, Index nObj_, Index *maxObjDim)
            {
                if (h) lexls_lse_destroy(h);
                h    = NULL;
                nVar = nVar_;
                nObj = nObj_;
                check(lexls_lse_create(&h, device, 1, nVar, nObj, maxObjDim));
                check(lexls_lse_round_layout(h, &lay)); // the small per-factorization arrays travel as ONE block each way
                round_in.assign(lay.in_bytes, 0);
                round_out.assign(lay.out_bytes, 0);
                reg_sent = false;
                cap = 0;
                for (Index k = 0; k < nObj; k++) cap += maxObjDim[k];
                dims.assign(nObj, 0);
                first_row.assign(nObj, 0);
                rank.assign(nObj, 0);
                first_col.assign(nObj, 0);
                x.resize(nVar);
                LOD.resize(cap, nVar + 1);
                PROBLEM_DATA.resize(cap, nVar + 1);
                FACTOR.resize(cap, nVar + 1);
                dWorkspace.resize(2 * std::max(cap, nVar) + nVar + 1);
                ctr_type.assign(cap, static_cast<uint8_t>(CTR_INACTIVE));
                fixed_idx.assign(nVar, 0);
                fixed_val.assign(nVar, 0.0);
                fixed_type.assign(nVar, static_cast<uint8_t>(CTR_ACTIVE_UB));
                nCtr = TotalRank = 0;
                check(lexls_lse_set_tolerance(h, parameters.tol_linear_dependence));
                check(lexls_lse_set_sensitivity_scan(h, sens_scan ? 1 : 0));
            }

            /// lexlse.h:1426-1442 (incl. initialize(), :1672-1693)
            void setObjDim(Index *ObjDim_)
            {
                nCtr = 0;
                for (Index k = 0; k < nObj; k++)
                {
                    dims[k]      = ObjDim_[k];
                    first_row[k] = nCtr;
                    nCtr += ObjDim_[k];
                    rank[k] = first_col[k] = 0;
                }
                nVarFixedInit = 0; // (the dimensions reach the device with the next factorize(): upload())
                TotalRank     = 0;
                ranks_fetched = total_rank_fetched = true;
                for (Index i = nVarFixed; i < nVar; i++) x(i) = 0.0;
            }

            /// lexlse.h:1467.  Every regularization type of typedefs.h:34-43 has a device path
            void setParameters(const ParametersLexLSE &p)
            {
                switch (p.regularization_type)
                {
                case REGULARIZATION_NONE:
                case REGULARIZATION_TIKHONOV:
                case REGULARIZATION_TIKHONOV_CG:
                case REGULARIZATION_R:
                case REGULARIZATION_R_NO_Z:
                case REGULARIZATION_RT_NO_Z:
                case REGULARIZATION_RT_NO_Z_CG:
                case REGULARIZATION_TIKHONOV_1:
                case REGULARIZATION_TIKHONOV_2:
                case REGULARIZATION_TEST: break;
                default: throw Exception("lexls_hip: unknown regularization type");
                }
                parameters = p;
                if (h) check(lexls_lse_set_tolerance(h, p.tol_linear_dependence));
            }
            /// lexlse.h:1477
            void setRegularizationFactor(Index ObjIndex, RealScalar factor)
            {
                if (reg_factor.size() < nObj) reg_factor.assign(nObj, 0.0);
                reg_factor[ObjIndex] = factor;
            }

            /// lexlse.h:1449-1462
            void setFixedVariablesCount(Index nVarFixed_)
            {
                if (nVarFixed_ > nVar) throw Exception("Cannot fix more than nVar variables");
                nVarFixed = nVarFixed_;
            }
            /// lexlse.h:1381-1388
            void fixVariable(Index VarIndex, RealScalar VarValue, ConstraintActivationType type = CTR_ACTIVE_UB)
            {
                fixed_idx[nVarFixedInit]  = VarIndex;
                fixed_val[nVarFixedInit]  = VarValue;
                fixed_type[nVarFixedInit] = static_cast<uint8_t>(type);
                x(nVarFixedInit)          = VarValue;
                nVarFixedInit++;
            }
            /// lexlse.h:1398-1419
            void fixVariables(Index nVarFixed_, Index *VarIndex, RealScalar *VarValue, ConstraintActivationType *type)
            {
                setFixedVariablesCount(nVarFixed_);
                nVarFixedInit = 0;
                for (Index k = 0; k < nVarFixed; k++) fixVariable(VarIndex[k], VarValue[k], type[k]);
            }

            /// lexlse.h:1511-1514
            void setProblem(const dMatrixConstRef &data)
            {
                for (Index j = 0; j < data.cols(); j++)
                    for (Index i = 0; i < data.rows(); i++) LOD(i, j) = data(i, j);
            }
            /// lexlse.h:1522-1530
            void setData(Index ObjIndex, const dMatrixConstRef &data)
            {
                if (ObjIndex >= nObj) throw Exception("ObjIndex >= nObj");
                for (Index j = 0; j <= nVar; j++)
                    for (Index i = 0; i < dims[ObjIndex]; i++) LOD(first_row[ObjIndex] + i, j) = data(i, j);
            }
            /// lexlse.h:1539-1543
            void setCtrStrided(Index CtrIndex, const RealScalar *row, Index stride, RealScalar rhs)
            {
                for (Index j = 0; j < nVar; j++) LOD(CtrIndex, j) = row[static_cast<size_t>(j) * stride];
                LOD(CtrIndex, nVar) = rhs;
            }
            void setCtr(Index CtrIndex, const RealScalar *row, RealScalar rhs) { setCtrStrided(CtrIndex, row, 1, rhs); }
            /// lexlse.h:1548-1552
            void setCtrType(Index ObjIndex, Index CtrIndex, ConstraintActivationType type) { ctr_type[first_row[ObjIndex] + CtrIndex] = static_cast<uint8_t>(type); }

            /// lexlse.h:117-506
            void factorize()
            {
                PROBLEM_DATA = LOD; // :119
                upload();
                check(lexls_lse_factorize(h));
                factor_on_host = false;
                ranks_fetched = total_rank_fetched = false; // fetched when asked for (getRank / getTotalRank) or with the solution
                lambda_pending = false;
            }
            /// lexlse.h:1015-1045
            void solve()
            {
                check(lexls_lse_solve(h));
                fetch_solution();
            }
            /// lexlse.h:1052-1131
            void solveLeastNorm_1()
            {
                check(lexls_lse_solve_least_norm(h));
                fetch_solution();
            }
            /// lexlse.h:1138-1213
            void solveLeastNorm_2()
            {
                check(lexls_lse_solve_least_norm_2(h));
                fetch_solution();
            }
            /// lexlse.h:1222-1277 (needs regularization_type TIKHONOV / TIKHONOV_2 / R, normally with all factors 0)
            void solveLeastNorm_3()
            {
                check(lexls_lse_solve_least_norm_3(h));
                fetch_solution();
            }

            /// lexlse.h:611-762; on return getWorkspace().head(nVarFixed + nLambda) = [lambda_fixed; lambda]
            bool ObjectiveSensitivity(Index ObjIndex, Index &CtrIndex2Remove, int &ObjIndex2Remove, RealScalar tol_wrong_sign_lambda,
                                      RealScalar tol_correct_sign_lambda, RealScalar &maxAbsValue)
            {
                if (ObjIndex >= nObj) throw Exception("ObjIndex >= nObj");
                check(lexls_lse_sensitivity(h, NULL, static_cast<int32_t>(ObjIndex), tol_wrong_sign_lambda, tol_correct_sign_lambda));
                check(lexls_lse_download_round(h, round_out.data(), NULL)); // verdict and largest violation in one copy
                int32_t s3[3];
                std::memcpy(s3, round_out.data() + lay.found, sizeof(s3));
                std::memcpy(&maxAbsValue, round_out.data() + lay.max_abs, sizeof(double));
                lambda_pending = true; // the multipliers themselves are fetched when getWorkspace() is read (getLambda, lexlsi.h:552-605)
                if (s3[0])
                {
                    CtrIndex2Remove = static_cast<Index>(s3[1]);
                    ObjIndex2Remove = s3[2];
                }
                return s3[0] != 0;
            }
            /// lexlse.h:511-602 (all wrong-sign multipliers, for deactivate_first_wrong_sign).  The multipliers come from the device
            /// (same kernel as the overload above); the scan of lexlse.h:866-910 runs here, in the reference's order (level ObjIndex,
            /// then ObjIndex-1 ... 0, then the fixed variables) and with its quirk for the fixed variables (it reads Lambda/ObjDim where
            /// LambdaFixed/nVarFixed are meant, lexlse.h:599-600; clipped to nVarFixed as in oracle/lexlse_oracle.h).  The
            /// CORRECT_SIGN_OF_LAMBDA marks of THIS overload then replace the ones the kernel made.
            void ObjectiveSensitivity(Index ObjIndex, RealScalar tol_wrong_sign_lambda, RealScalar tol_correct_sign_lambda,
                                      std::vector<ConstraintInfo> &ctr_wrong_sign)
            {
                if (ObjIndex >= nObj) throw Exception("ObjIndex >= nObj");
                check(lexls_lse_sensitivity(h, NULL, static_cast<int32_t>(ObjIndex), tol_wrong_sign_lambda, tol_correct_sign_lambda));
                std::vector<double> lam(nVar + cap);
                check(lexls_lse_get_lambda(h, lam.data()));
                for (Index i = 0; i < nVar + cap && i < dWorkspace.size(); i++) dWorkspace(i) = lam[i];
                lambda_pending       = false;
                const double *Lambda = lam.data() + nVarFixed;
                auto scan = [&](int obj, bool fixed, Index first, Index count) {
                    for (Index k = 0; k < count; k++)
                    {
                        const Index ind = fixed ? k : first + k;
                        uint8_t &type   = fixed ? fixed_type[ind] : ctr_type[ind];
                        if (type == CTR_ACTIVE_EQ || type == CORRECT_SIGN_OF_LAMBDA) continue;
                        double a = Lambda[ind];
                        if (type == CTR_ACTIVE_LB) a = -a;
                        if (a > tol_correct_sign_lambda)
                            type = static_cast<uint8_t>(CORRECT_SIGN_OF_LAMBDA);
                        else if (a < -tol_wrong_sign_lambda)
                            ctr_wrong_sign.push_back(ConstraintInfo(obj, static_cast<int>(k)));
                    }
                };
                for (Index k = ObjIndex + 1; k--;) scan(static_cast<int>(k), false, first_row[k], dims[k]);
                if (nVarFixed > 0)
                {
                    scan(-1, true, 0, std::min(dims[0], nVarFixed));
                    check(lexls_lse_set_fixed_type(h, fixed_type.data()));
                }
                check(lexls_lse_set_ctr_type(h, ctr_type.data()));
            }

            /// lexlse.h:1560-1582
            dVectorType &get_v()
            {
                check(lexls_lse_residual(h));
                std::vector<double> v(cap);
                check(lexls_lse_get_v(h, v.data()));
                lambda_pending = false; // the residuals take the workspace, as in the reference
                for (Index i = 0; i < nCtr; i++) dWorkspace(i) = v[i];
                return dWorkspace;
            }

            const dVectorType &get_x() const { return x; }
            Index getDim(Index k) const { return dims[k]; }
            Index getRank(Index k) const
            {
                fetch_ranks();
                return rank[k];
            }
            Index get_nObj() const { return nObj; }
            Index get_nVar() const { return nVar; }
            Index getTotalRank() const
            {
                if (!total_rank_fetched) fetch_ranks();
                return TotalRank;
            }
            Index getFixedVariablesCount() const { return nVarFixed; }
            /// lexlse.h:1495.  The indices as given to fixVariable(); the reference hands out its working copy, which factorize()
            /// rewrites by the chained-index rule (:146-153) — nothing in the reference reads it afterwards.
            iVectorType getFixedVarIndex() const
            {
                iVectorType v(nVarFixed);
                for (Index i = 0; i < nVarFixed; i++) v(i) = fixed_idx[i];
                return v;
            }
            /// lexlse.h:1636-1650: the by-products of REGULARIZATION_TIKHONOV_1 (regularize_tikhonov_1_test); other types leave them unset
            const dMatrixType &get_X_mu()
            {
                fetch_mu();
                return X_mu;
            }
            const dMatrixType &get_X_mu_rhs()
            {
                fetch_mu();
                return X_mu_rhs;
            }
            const dVectorType &get_residual_mu()
            {
                fetch_mu();
                return residual_mu;
            }
            /// lexlse.h:770-861: the overload "to form the matrix of Lagrange multipliers (for debugging purposes)": the multipliers of objective
            /// ObjIndex, from the residual of the factorization, into the workspace ([lambda_fixed; lambda], read them with getWorkspace()) — no
            /// removal decision, no CORRECT_SIGN_OF_LAMBDA marks.  Served by the same device routine as the deciding overload with tolerances no
            /// multiplier can meet (it neither marks nor finds a candidate) and the scan over the following objectives switched off for the call.
            /// (The reference's by-product for the experimental regularization type 7 — residual_mu instead of the factorization's residual,
            /// :800-811 — is not provided.)
            void ObjectiveSensitivity(Index ObjIndex)
            {
                if (ObjIndex >= nObj) throw Exception("ObjIndex >= nObj");
                const double never = std::numeric_limits<double>::infinity();
                if (sens_scan) check(lexls_lse_set_sensitivity_scan(h, 0));
                const int rc = lexls_lse_sensitivity(h, NULL, static_cast<int32_t>(ObjIndex), never, never);
                if (sens_scan) check(lexls_lse_set_sensitivity_scan(h, 1));
                check(rc);
                lambda_pending = true;
            }
            const dVectorType &getWorkspace()
            {
                if (lambda_pending)
                {
                    std::vector<double> lam(nVar + cap);
                    check(lexls_lse_get_lambda(h, lam.data()));
                    for (Index i = 0; i < nVar + cap && i < dWorkspace.size(); i++) dWorkspace(i) = lam[i];
                    lambda_pending = false;
                }
                return dWorkspace;
            }
            const dMatrixType &get_data() const { return PROBLEM_DATA; }
            const dMatrixType &get_lexqr()
            {
                if (!factor_on_host)
                {
                    check(lexls_lse_get_factor(h, FACTOR.data()));
                    factor_on_host = true;
                }
                return FACTOR;
            }
            const char *last_kernel() const { return lexls_lse_last_kernel(h); }

            /// lexlse.h:1654-1658
            void reset()
            {
                nVarFixedInit = 0;
                TotalRank     = 0;
                x.setZero();
            }

        private:
            static void check(int rc)
            {
                if (rc != LEXLS_OK) throw Exception(std::string("liblexls_hip: ") + lexls_last_error());
            }
            /// what setObjDim / fixVariable / setCtrType collected goes to the device as one block (lexls_lse_upload_round), the matrix as a
            /// second copy; the regularization setup only when it changed
            void upload()
            {
                unsigned char *in = round_in.data();
                const uint32_t nf = nVarFixed;
                std::memcpy(in + lay.dims, dims.data(), sizeof(uint32_t) * nObj);
                std::memcpy(in + lay.nfixed, &nf, sizeof(uint32_t));
                std::memcpy(in + lay.fixed_idx, fixed_idx.data(), sizeof(uint32_t) * nVar);
                std::memcpy(in + lay.fixed_val, fixed_val.data(), sizeof(double) * nVar);
                in[lay.skip] = 0;
                const int32_t none = -1;
                std::memcpy(in + lay.obj_index, &none, sizeof(int32_t));
                std::memcpy(in + lay.fixed_type, fixed_type.data(), nVar);
                std::memcpy(in + lay.ctr_type, ctr_type.data(), cap);
                check(lexls_lse_upload_round(h, in, 0));
                check(lexls_lse_set_problem_host(h, LOD.data()));
                if (reg_factor.size() < nObj) reg_factor.assign(nObj, 0.0);
                const int type = static_cast<int>(parameters.regularization_type);
                if (!reg_sent || type != sent_type || parameters.variable_regularization_factor != sent_variable ||
                    parameters.max_number_of_CG_iterations != sent_cg || (type != 0 && reg_factor != sent_factor))
                {
                    check(lexls_lse_set_cg_iterations(h, parameters.max_number_of_CG_iterations));
                    check(lexls_lse_set_regularization(h, type, reg_factor.data(), 0, parameters.variable_regularization_factor));
                    reg_sent      = true;
                    sent_type     = type;
                    sent_variable = parameters.variable_regularization_factor;
                    sent_cg       = parameters.max_number_of_CG_iterations;
                    sent_factor   = reg_factor;
                }
            }
            /// x and the total rank in one copy (the out slab of the round block)
            void fetch_solution()
            {
                check(lexls_lse_download_round(h, round_out.data(), NULL));
                std::memcpy(x.data(), round_out.data() + lay.x, sizeof(double) * nVar);
                uint32_t tr;
                std::memcpy(&tr, round_out.data() + lay.total_rank, sizeof(uint32_t));
                TotalRank          = tr;
                total_rank_fetched = true;
            }
            void fetch_ranks() const
            {
                if (ranks_fetched) return;
                Index tr = 0;
                check(lexls_lse_get_ranks(h, rank.data(), first_col.data(), &tr));
                TotalRank     = tr;
                ranks_fetched = total_rank_fetched = true;
            }

            void fetch_mu()
            {
                if (parameters.regularization_type != REGULARIZATION_TIKHONOV_1)
                    throw Exception("lexls_hip: X_mu, X_mu_rhs and residual_mu are produced by REGULARIZATION_TIKHONOV_1 only");
                X_mu.resize(nVar, nObj);
                X_mu_rhs.resize(nVar, nObj);
                residual_mu.resize(cap);
                check(lexls_lse_get_mu(h, X_mu.data(), X_mu_rhs.data(), residual_mu.data()));
            }

            lexls_lse_t h;
            Index nVar, nObj, nCtr, cap, nVarFixed, nVarFixedInit;
            mutable Index TotalRank;
            int device;
            bool factor_on_host = false;
            bool sens_scan      = false;
            lexls_round_layout lay;
            std::vector<unsigned char> round_in, round_out;
            mutable bool ranks_fetched = true, total_rank_fetched = true;
            bool lambda_pending = false, reg_sent = false;
            int sent_type = 0;
            double sent_variable = 0.0;
            Index sent_cg = 0;
            std::vector<double> sent_factor;
            ParametersLexLSE parameters;
            std::vector<Index> dims, first_row, fixed_idx;
            mutable std::vector<Index> rank, first_col;
            std::vector<double> fixed_val, reg_factor;
            std::vector<uint8_t> fixed_type, ctr_type;
            dMatrixType LOD, PROBLEM_DATA, FACTOR, X_mu, X_mu_rhs;
            dVectorType x, dWorkspace, residual_mu;
        };
    } // namespace internal

    /// public wrapper of the reference (include/lexls/lexls.h:16-68)
    class LexLSE
    {
    public:
        LexLSE() {}
        LexLSE(Index nVar_, Index nObj_, Index *ObjDim_)
        {
            resize(nVar_, nObj_, ObjDim_);
            setObjDim(ObjDim_);
        }
        void resize(Index nVar_, Index nObj_, Index *ObjDim_) { lexlse.resize(nVar_, nObj_, ObjDim_); }
        void setObjDim(Index *ObjDim_) { lexlse.setObjDim(ObjDim_); }
        /// the reference's wrapper exposes no data setter (lexls.h:16-68); this one forwards setData
        void setData(Index ObjIndex, const dMatrixConstRef &data) { lexlse.setData(ObjIndex, data); }
        const dVectorType &solve(Index solve_option = 0)
        {
            lexlse.factorize();
            switch (solve_option)
            {
            case 0: lexlse.solve(); break;
            case 1: lexlse.solveLeastNorm_1(); break;
            case 2: lexlse.solveLeastNorm_2(); break;
            case 3: lexlse.solveLeastNorm_3(); break;
            default: break;
            }
            return lexlse.get_x();
        }

    private:
        internal::LexLSE lexlse;
    };
} // namespace LexLS
