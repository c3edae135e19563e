/*
 * Derived work: this header restates, for a different equality-solver back end, host-side interface and control flow of
 * jrl-umi3218/lexls (include/lexls/objective.h), whose notice is retained as its BSD 3-clause licence requires:
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
// One objective (priority level) of a LexLSI problem: data, working set, residual bookkeeping.
// Host side by design (north star: "LexLSI's outer active-set loop is kept on the host").
//
// Restates the behaviour of the reference include/lexls/objective.h; each method cites the lines
// it follows.  Dense products are one left-to-right fma chain per row (the reference leaves the
// order to Eigen's GEMV).
#pragma once
#include <type_traits>
#include <utility>

#include <lexls/typedefs.h>
#include <lexls/workingset.h>

namespace LexLS
{
    namespace internal
    {
        /// detects LSE::setCtrIndexed(Index row, size_t first_element, Index leading_dimension, unsigned use_ub) -> bool
        template <class LSE>
        struct has_setCtrIndexed
        {
            template <class T>
            static auto test(int) -> decltype(std::declval<T &>().setCtrIndexed(Index(), size_t(), Index(), 0u), std::true_type());
            template <class>
            static std::false_type test(...);
            typedef decltype(test<LSE>(0)) type;
        };
        template <class LSE>
        inline bool try_setCtrIndexed(LSE &l, Index row, size_t first_element, Index ld, unsigned use_ub, std::true_type)
        {
            return l.setCtrIndexed(row, first_element, ld, use_ub);
        }
        template <class LSE>
        inline bool try_setCtrIndexed(LSE &, Index, size_t, Index, unsigned, std::false_type)
        {
            return false;
        }

        class Objective
        {
        public:
            Objective() : nVar(0), nCtr(0), lb_index(0), ub_index(0), obj_type(GENERAL_OBJECTIVE), regularization_factor(0.0), v0_is_specified(false) {}

            /// objective.h:36-71
            void resize(Index nCtr_, Index nVar_, ObjectiveType obj_type_)
            {
                obj_type = obj_type_;
                nCtr     = nCtr_;
                nVar     = nVar_;
                working_set.resize(nCtr);
                v.resize(nCtr);
                dv.resize(nCtr);
                Ax.resize(nCtr);
                Adx.resize(nCtr);
                if (obj_type == GENERAL_OBJECTIVE)
                {
                    data.resize(nCtr, nVar + 2); // [A, lb, ub]
                    lb_index = nVar;
                    ub_index = nVar + 1;
                }
                else if (obj_type == SIMPLE_BOUNDS_OBJECTIVE)
                {
                    data.resize(nCtr, 2); // [lb, ub]
                    var_index.resize(nCtr);
                    lb_index = 0;
                    ub_index = 1;
                }
                else
                {
                    throw Exception("Unknown objective type");
                }
                v.setZero();
                dv.setZero();
            }

            /// objective.h:73-103
            void ensureZeroCtrViolationForSimpleBounds(dVectorType &x)
            {
                if (obj_type != SIMPLE_BOUNDS_OBJECTIVE) return;
                for (Index c = 0; c < nCtr; c++)
                {
                    const Index var = getVarIndex(c);
                    switch (getCtrType(c))
                    {
                    case CTR_INACTIVE:
                        x(var) = 0.5 * (data(c, lb_index) + data(c, ub_index));
                        break;
                    case CTR_ACTIVE_EQ:
                    case CTR_ACTIVE_UB:
                        x(var) = data(c, ub_index);
                        break;
                    case CTR_ACTIVE_LB:
                        x(var) = data(c, lb_index);
                        break;
                    default:
                        break;
                    }
                }
            }

            /// objective.h:115-172 (hot-start working-set repair; needs Ax)
            void formInitialWorkingSet(dVectorType &x, bool modify_type_active_enabled, bool modify_type_inactive_enabled, bool modify_x_guess_enabled)
            {
                if (modify_type_active_enabled || modify_type_inactive_enabled)
                {
                    for (Index c = 0; c < nCtr; c++)
                    {
                        if (!isActive(c) && modify_type_inactive_enabled)
                        {
                            if (Ax(c) <= data(c, lb_index))
                                activate(c, CTR_ACTIVE_LB);
                            else if (Ax(c) >= data(c, ub_index))
                                activate(c, CTR_ACTIVE_UB);
                        }
                        else if (getCtrType(c) == CTR_ACTIVE_LB && modify_type_active_enabled)
                        {
                            if (Ax(c) > data(c, lb_index))
                            {
                                deactivate(working_set.getCtrIndex(c));
                                if (Ax(c) >= data(c, ub_index)) activate(c, CTR_ACTIVE_UB);
                            }
                        }
                        else if (getCtrType(c) == CTR_ACTIVE_UB && modify_type_active_enabled)
                        {
                            if (Ax(c) < data(c, ub_index))
                            {
                                deactivate(working_set.getCtrIndex(c));
                                if (Ax(c) <= data(c, lb_index)) activate(c, CTR_ACTIVE_LB);
                            }
                        }
                    }
                }
                if (getObjType() == SIMPLE_BOUNDS_OBJECTIVE && modify_x_guess_enabled)
                {
                    ensureZeroCtrViolationForSimpleBounds(x);
                    initialize_Ax(x);
                }
            }

            /// objective.h:183-237
            void initialize_v0(RealScalar tol_feasibility, bool set_min_init_ctr_violation)
            {
                for (Index c = 0; c < nCtr; c++) v(c) = Ax(c) - 0.5 * (data(c, lb_index) + data(c, ub_index));

                for (Index k = 0; k < getActiveCtrCount(); k++)
                {
                    const Index c                       = getActiveCtrIndex(k);
                    const ConstraintActivationType type = getActiveCtrType(k);
                    if (type == CTR_ACTIVE_LB)
                        v(c) = Ax(c) - data(c, lb_index);
                    else if (type == CTR_ACTIVE_UB)
                        v(c) = Ax(c) - data(c, ub_index);
                }

                for (Index c = 0; c < nCtr; c++)
                {
                    if (isActive(c)) continue;
                    if (set_min_init_ctr_violation)
                    {
                        if (Ax(c) <= data(c, lb_index))
                            v(c) = Ax(c) - data(c, lb_index);
                        else if (Ax(c) >= data(c, ub_index))
                            v(c) = Ax(c) - data(c, ub_index);
                        else
                            v(c) = 0.0;
                    }
                    else if (Ax(c) >= data(c, lb_index) - tol_feasibility && Ax(c) <= data(c, ub_index) + tol_feasibility)
                    {
                        v(c) = 0.0;
                    }
                }
            }

            /// objective.h:242-255
            void initialize_Ax(const dVectorType &x) { apply_A(x, Ax); }
            /// objective.h:260-273
            void form_Adx(const dVectorType &dx) { apply_A(dx, Adx); }

            /// objective.h:288-338: dv = -v, and for active rows dv += (Ax + Adx - rhs)
            void formStep(const dVectorType &dx)
            {
                form_Adx(dx);
                for (Index c = 0; c < nCtr; c++) dv(c) = -v(c);
                for (Index k = 0; k < getActiveCtrCount(); k++)
                {
                    const Index c = getActiveCtrIndex(k);
                    RealScalar rhs;
                    switch (getActiveCtrType(k))
                    {
                    case CTR_ACTIVE_EQ:
                    case CTR_ACTIVE_UB:
                        rhs = data(c, ub_index);
                        break;
                    case CTR_ACTIVE_LB:
                        rhs = data(c, lb_index);
                        break;
                    default:
                        throw Exception("UNKNOWN constraint type");
                    }
                    dv(c) += Ax(c) + Adx(c) - rhs;
                }
            }

            /// objective.h:353-374
            void phase1(dVectorType &x, bool x_guess_is_specified, bool modify_type_active_enabled, bool modify_type_inactive_enabled,
                        bool modify_x_guess_enabled, bool set_min_init_ctr_violation, RealScalar tol_feasibility)
            {
                initialize_Ax(x);
                if (!v0_is_specified)
                {
                    if (x_guess_is_specified)
                        formInitialWorkingSet(x, modify_type_active_enabled, modify_type_inactive_enabled, modify_x_guess_enabled);
                    initialize_v0(tol_feasibility, set_min_init_ctr_violation);
                }
            }

            /// objective.h:382-390
            void activate(Index CtrIndex, ConstraintActivationType type)
            {
                if (CtrIndex >= nCtr) throw Exception("CtrIndex >= nCtr");
                working_set.activate(CtrIndex, type);
            }

            /// objective.h:398-406
            void deactivate(Index CtrIndexActive)
            {
                if (CtrIndexActive >= getActiveCtrCount()) throw Exception("CtrIndexActive >= number of active constraints");
                working_set.deactivate(CtrIndexActive);
            }

            /// objective.h:434-494: gather the active rows into the equality solver.
            /// Active RHS convention: EQ/UB -> ub, LB -> lb.
            template <class LSE>
            void formLexLSE(LSE &lexlse, Index &counter, Index ObjIndex)
            {
                if (obj_type == SIMPLE_BOUNDS_OBJECTIVE)
                {
                    lexlse.setFixedVariablesCount(getActiveCtrCount());
                    for (Index k = 0; k < getActiveCtrCount(); k++)
                    {
                        const Index c                       = getActiveCtrIndex(k);
                        const Index var                     = getVarIndex(c);
                        const ConstraintActivationType type = getActiveCtrType(k);
                        if (type == CTR_ACTIVE_LB)
                            lexlse.fixVariable(var, data(c, 0), CTR_ACTIVE_LB);
                        else if (type == CTR_ACTIVE_UB)
                            lexlse.fixVariable(var, data(c, 1), CTR_ACTIVE_UB);
                        else if (type == CTR_ACTIVE_EQ)
                            lexlse.fixVariable(var, data(c, 1), CTR_ACTIVE_EQ);
                    }
                }
                else
                {
                    for (Index k = 0; k < getActiveCtrCount(); k++)
                    {
                        const Index c                       = getActiveCtrIndex(k);
                        const ConstraintActivationType type = getActiveCtrType(k);
                        RealScalar rhs                      = 0;
                        if (type == CTR_ACTIVE_EQ || type == CTR_ACTIVE_UB)
                            rhs = data(c, nVar + 1);
                        else if (type == CTR_ACTIVE_LB)
                            rhs = data(c, nVar);
                        lexlse.setCtrType(ObjIndex, k, type);
                        // a backend that keeps the constraint data resident (batched device path) only needs
                        // to know WHICH row and which bound; everybody else gets the numbers
                        if (!try_setCtrIndexed(lexlse, counter, data_offset + c, data.rows(), type == CTR_ACTIVE_LB ? 0u : 1u,
                                               typename has_setCtrIndexed<LSE>::type()))
                            lexlse.setCtrStrided(counter, &data(c, 0), data.rows(), rhs);
                        counter++;
                    }
                    lexlse.setRegularizationFactor(ObjIndex, regularization_factor);
                }
            }

            /// objective.h:521-578: ratio test over the INACTIVE constraints in their current
            /// (history dependent) order; strict '<' keeps the first minimiser.
            bool checkBlockingConstraints(Index &CtrIndexBlocking, ConstraintActivationType &CtrTypeBlocking, RealScalar &alpha,
                                          RealScalar tol_feasibility) const
            {
                const RealScalar alpha_input = alpha;
                for (Index k = 0; k < getInactiveCtrCount(); k++)
                {
                    const Index c        = getInactiveCtrIndex(k);
                    const RealScalar den = Adx(c) - dv(c);
                    ConstraintActivationType type;
                    RealScalar rhs;
                    if (den < -tol_feasibility)
                    {
                        type = CTR_ACTIVE_LB;
                        rhs  = data(c, lb_index);
                    }
                    else if (den > tol_feasibility)
                    {
                        type = CTR_ACTIVE_UB;
                        rhs  = data(c, ub_index);
                    }
                    else
                    {
                        continue;
                    }
                    const RealScalar num = rhs - Ax(c) + v(c);
                    RealScalar ratio     = num / den;
                    if (ratio < 0) ratio = 0;
                    if (ratio < alpha)
                    {
                        alpha            = ratio;
                        CtrIndexBlocking = c;
                        CtrTypeBlocking  = type;
                    }
                }
                return alpha < alpha_input;
            }

            /// objective.h:585-589
            void step(RealScalar alpha)
            {
                for (Index c = 0; c < nCtr; c++)
                {
                    v(c) += alpha * dv(c);
                    Ax(c) += alpha * Adx(c);
                }
            }

            const dVectorType &get_v() const { return v; }
            const dVectorType &get_dv() const { return dv; }
            const dVectorType &get_Ax() const { return Ax; }

            /// objective.h:611-630
            void getConstraintViolation(dVectorType &ctr_violation) const
            {
                ctr_violation.resize(nCtr);
                for (Index c = 0; c < nCtr; c++)
                {
                    if (Ax(c) <= data(c, lb_index))
                        ctr_violation(c) = Ax(c) - data(c, lb_index);
                    else if (Ax(c) >= data(c, ub_index))
                        ctr_violation(c) = Ax(c) - data(c, ub_index);
                    else
                        ctr_violation(c) = 0.0;
                }
            }

            void resetActiveSet() { working_set.reset(); }
            Index getActiveCtrCount() const { return working_set.getActiveCtrCount(); }
            Index getActiveCtrIndex(Index k) const { return working_set.getActiveCtrIndex(k); }
            Index getCtrIndex(Index k) const { return working_set.getCtrIndex(k); }
            ConstraintActivationType getActiveCtrType(Index k) const { return working_set.getActiveCtrType(k); }
            ConstraintActivationType getCtrType(Index k) const { return working_set.getCtrType(k); }
            Index getInactiveCtrCount() const { return working_set.getInactiveCtrCount(); }
            Index getInactiveCtrIndex(Index k) const { return working_set.getInactiveCtrIndex(k); }
            Index getVarIndex(Index k) const { return var_index(k); }
            ObjectiveType getObjType() const { return obj_type; }
            Index getDim() const { return nCtr; }
            const dMatrixType &getData() const { return data; }
            bool isActive(Index CtrIndex) const { return working_set.isActive(CtrIndex); }
            bool getFlag_v0_is_specified() const { return v0_is_specified; }
            void setFlag_v0_is_specified(bool flag) { v0_is_specified = flag; }

            /// objective.h:765-769
            void set_v0(const dVectorType &v_)
            {
                v = v_;
                setFlag_v0_is_specified(true);
            }

            /// objective.h:774-788
            void relax_bounds(Index CtrIndex, ConstraintActivationType CtrType, RealScalar p)
            {
                if (CtrType == CTR_ACTIVE_LB)
                    data(CtrIndex, lb_index) -= p;
                else if (CtrType == CTR_ACTIVE_UB)
                    data(CtrIndex, ub_index) += p;
                else
                    throw Exception("Should not be here");
            }

            /// objective.h:793-815
            void setData(const dMatrixConstRef &data_) { data = data_; }
            /// position of this objective's [A | lb | ub] block (column-major, first element) inside the caller's flat data of
            /// the problem; only meaningful to backends that gather rows from a resident copy of that flat data
            void setDataOffset(size_t off) { data_offset = off; }
            void setData(const Index *var_index_, const dMatrixConstRef &data_)
            {
                for (Index k = 0; k < nCtr; k++) var_index(k) = var_index_[k];
                data = data_;
            }
            void setData(Index k, Index var_index_, RealScalar lb_, RealScalar ub_)
            {
                var_index(k) = var_index_;
                data(k, 0)   = lb_;
                data(k, 1)   = ub_;
            }

            /// objective.h:820-837
            void setRegularization(RealScalar factor)
            {
                if (obj_type == SIMPLE_BOUNDS_OBJECTIVE)
                    printf("WARNING: setting a nonzero regularization factor has no effect on an objective of type SIMPLE_BOUNDS_OBJECTIVE. \n");
                regularization_factor = factor;
            }
            RealScalar getRegularization() const { return regularization_factor; }

            /// objective.h:845-857
            bool isZeroNormal(Index CtrIndex) const
            {
                if (obj_type != GENERAL_OBJECTIVE) return false;
                RealScalar s = 0.0;
                for (Index j = 0; j < nVar; j++) s = std::fma(data(CtrIndex, j), data(CtrIndex, j), s);
                return s == 0.0;
            }

        private:
            /// out = A*x for general objectives, out(k) = x(var_index(k)) for simple bounds
            void apply_A(const dVectorType &x, dVectorType &out) const
            {
                if (obj_type == GENERAL_OBJECTIVE)
                {
                    // per row i the chain s = fma(A(i,j), x(j), s) over ascending j; the loops are interchanged (the data are column-major:
                    // contiguous in i), which leaves every row's chain — and its result — as it is
                    if (nCtr == 0) return;
                    for (Index i = 0; i < nCtr; i++) out(i) = 0.0;
                    for (Index j = 0; j < nVar; j++)
                    {
                        const RealScalar xj    = x(j);
                        const RealScalar *col  = &data(0, j);
                        RealScalar *o          = &out(0);
                        for (Index i = 0; i < nCtr; i++) o[i] = std::fma(col[i], xj, o[i]);
                    }
                }
                else
                {
                    for (Index k = 0; k < nCtr; k++) out(k) = x(var_index(k));
                }
            }

            Index nVar;
            Index nCtr;
            Index lb_index;
            Index ub_index;
            ObjectiveType obj_type;
            iVectorType var_index;
            dMatrixType data;
            size_t data_offset = 0;
            WorkingSet working_set;
            dVectorType v;
            dVectorType dv;
            dVectorType Ax;
            dVectorType Adx;
            RealScalar regularization_factor;
            bool v0_is_specified;
        };
    } // namespace internal
} // namespace LexLS
