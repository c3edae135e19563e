/*
 * Derived work: this header restates, for a different equality-solver back end, host-side interface and control flow of
 * jrl-umi3218/lexls (include/lexls/typedefs.h), whose notice is retained as its BSD 3-clause licence requires:
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
// Host-side types of the lexls-mi355x drop-in.
//
// Mirrors the *interface* of the reference's include/lexls/typedefs.h (enums :32-76, parameter
// structs :78-295, Exception :300-314, ConstraintInfo :323-375, WorkingSetLogEntry :380-432,
// ConstraintIdentifier :439-527, internal::ObjectiveInfo :621-670) so that code written against the
// reference compiles against these headers.  The reference stores everything in Eigen types; Eigen is
// not a dependency here: dMatrixType / dVectorType are small owning column-major containers and
// dMatrixConstRef is a non-owning (ptr, rows, cols, ld) view, which is also what crosses the C ABI.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

namespace LexLS
{
    typedef unsigned int Index;
    typedef double RealScalar;

    // ---------------------------------------------------------------------------------------------
    // minimal dense containers (column-major, like the reference's dMatrixType, typedefs.h:19)
    // ---------------------------------------------------------------------------------------------
    class dVectorType
    {
    public:
        dVectorType() {}
        explicit dVectorType(Index n) : d(n, 0.0) {}
        dVectorType(const RealScalar *p, Index n) : d(p, p + n) {}
        void resize(Index n) { d.assign(n, 0.0); }
        void setZero() { std::fill(d.begin(), d.end(), 0.0); }
        Index size() const { return static_cast<Index>(d.size()); }
        RealScalar &operator()(Index i) { return d[i]; }
        const RealScalar &operator()(Index i) const { return d[i]; }
        RealScalar &operator[](Index i) { return d[i]; }
        const RealScalar &operator[](Index i) const { return d[i]; }
        RealScalar &coeffRef(Index i) { return d[i]; }
        const RealScalar &coeff(Index i) const { return d[i]; }
        RealScalar *data() { return d.data(); }
        const RealScalar *data() const { return d.data(); }

    private:
        std::vector<RealScalar> d;
    };

    class iVectorType
    {
    public:
        iVectorType() {}
        explicit iVectorType(Index n) : d(n, 0) {}
        void resize(Index n) { d.assign(n, 0); }
        void setZero() { std::fill(d.begin(), d.end(), 0u); }
        Index size() const { return static_cast<Index>(d.size()); }
        Index &operator()(Index i) { return d[i]; }
        const Index &operator()(Index i) const { return d[i]; }
        Index &operator[](Index i) { return d[i]; }
        const Index &operator[](Index i) const { return d[i]; }
        Index &coeffRef(Index i) { return d[i]; }
        const Index &coeff(Index i) const { return d[i]; }
        Index *data() { return d.data(); }
        const Index *data() const { return d.data(); }

    private:
        std::vector<Index> d;
    };

    /// non-owning column-major view (what `const dMatrixConstRef&` is in the reference's signatures)
    struct dMatrixConstRef
    {
        const RealScalar *ptr;
        Index nrows, ncols, ld;
        dMatrixConstRef(const RealScalar *p, Index r, Index c, Index ld_) : ptr(p), nrows(r), ncols(c), ld(ld_) {}
        dMatrixConstRef(const RealScalar *p, Index r, Index c) : ptr(p), nrows(r), ncols(c), ld(r) {}
        Index rows() const { return nrows; }
        Index cols() const { return ncols; }
        const RealScalar &operator()(Index i, Index j) const { return ptr[i + static_cast<size_t>(j) * ld]; }
        const RealScalar &coeffRef(Index i, Index j) const { return (*this)(i, j); }
    };

    class dMatrixType
    {
    public:
        dMatrixType() : nrows(0), ncols(0) {}
        dMatrixType(Index r, Index c) : nrows(r), ncols(c), d(static_cast<size_t>(r) * c, 0.0) {}
        void resize(Index r, Index c)
        {
            nrows = r;
            ncols = c;
            d.assign(static_cast<size_t>(r) * c, 0.0);
        }
        void setZero() { std::fill(d.begin(), d.end(), 0.0); }
        Index rows() const { return nrows; }
        Index cols() const { return ncols; }
        RealScalar &operator()(Index i, Index j) { return d[i + static_cast<size_t>(j) * nrows]; }
        const RealScalar &operator()(Index i, Index j) const { return d[i + static_cast<size_t>(j) * nrows]; }
        RealScalar &coeffRef(Index i, Index j) { return (*this)(i, j); }
        const RealScalar &coeff(Index i, Index j) const { return (*this)(i, j); }
        RealScalar *data() { return d.data(); }
        const RealScalar *data() const { return d.data(); }
        operator dMatrixConstRef() const { return dMatrixConstRef(d.data(), nrows, ncols, nrows); }
        dMatrixType &operator=(const dMatrixConstRef &m)
        {
            resize(m.rows(), m.cols());
            for (Index j = 0; j < ncols; j++)
                for (Index i = 0; i < nrows; i++) (*this)(i, j) = m(i, j);
            return *this;
        }

    private:
        Index nrows, ncols;
        std::vector<RealScalar> d;
    };

    // ---------------------------------------------------------------------------------------------
    // enums: same names and numeric values as the reference (typedefs.h:32-76)
    // ---------------------------------------------------------------------------------------------
    enum RegularizationType
    {
        REGULARIZATION_NONE = 0,
        REGULARIZATION_TIKHONOV,
        REGULARIZATION_TIKHONOV_CG,
        REGULARIZATION_R,
        REGULARIZATION_R_NO_Z,
        REGULARIZATION_RT_NO_Z,
        REGULARIZATION_RT_NO_Z_CG,
        REGULARIZATION_TIKHONOV_1,
        REGULARIZATION_TIKHONOV_2,
        REGULARIZATION_TEST
    };

    enum TerminationStatus
    {
        TERMINATION_STATUS_UNKNOWN = -1,
        PROBLEM_SOLVED,
        PROBLEM_SOLVED_CYCLING_HANDLING,
        MAX_NUMBER_OF_FACTORIZATIONS_EXCEEDED
    };

    enum ObjectiveType
    {
        GENERAL_OBJECTIVE = 0,
        SIMPLE_BOUNDS_OBJECTIVE
    };

    enum ConstraintActivationType
    {
        CTR_INACTIVE = 0,
        CTR_ACTIVE_LB,
        CTR_ACTIVE_UB,
        CTR_ACTIVE_EQ,
        CORRECT_SIGN_OF_LAMBDA
    };

    /// reference typedefs.h:78-125 (only REGULARIZATION_NONE is implemented on the device path)
    class ParametersLexLSE
    {
    public:
        RealScalar tol_linear_dependence;
        Index max_number_of_CG_iterations;
        RegularizationType regularization_type;
        RealScalar variable_regularization_factor;

        ParametersLexLSE() { setDefaults(); }
        void setDefaults()
        {
            tol_linear_dependence          = 1e-12;
            max_number_of_CG_iterations    = 10;
            regularization_type            = REGULARIZATION_NONE;
            variable_regularization_factor = 0.0;
        }
    };

    /// reference typedefs.h:127-295
    class ParametersLexLSI
    {
    public:
        Index max_number_of_factorizations;
        RealScalar tol_linear_dependence;
        RealScalar tol_wrong_sign_lambda;
        RealScalar tol_correct_sign_lambda;
        RealScalar tol_feasibility;
        RegularizationType regularization_type;
        Index max_number_of_CG_iterations;
        RealScalar variable_regularization_factor;
        bool cycling_handling_enabled;
        Index cycling_max_counter;
        RealScalar cycling_relax_step;
        std::string output_file_name;
        bool modify_x_guess_enabled;
        bool modify_type_active_enabled;
        bool modify_type_inactive_enabled;
        bool set_min_init_ctr_violation;
        bool use_phase1_v0;
        bool log_working_set_enabled;
        bool deactivate_first_wrong_sign;

        ParametersLexLSI() { setDefaults(); }
        void setDefaults()
        {
            max_number_of_factorizations   = 200;
            tol_linear_dependence          = 1e-12;
            tol_wrong_sign_lambda          = 1e-08;
            tol_correct_sign_lambda        = 1e-12;
            tol_feasibility                = 1e-13;
            cycling_handling_enabled       = false;
            cycling_max_counter            = 50;
            cycling_relax_step             = 1e-08;
            regularization_type            = REGULARIZATION_NONE;
            max_number_of_CG_iterations    = 10;
            variable_regularization_factor = 0.0;
            modify_x_guess_enabled         = false;
            modify_type_active_enabled     = false;
            modify_type_inactive_enabled   = false;
            set_min_init_ctr_violation     = true;
            use_phase1_v0                  = false;
            log_working_set_enabled        = false;
            deactivate_first_wrong_sign    = false;
        }
    };

    /// reference typedefs.h:300-314
    class Exception : public std::exception
    {
    public:
        explicit Exception(const char *message) : msg(message) {}
        explicit Exception(const std::string &message) : msg(message) {}
        ~Exception() throw() {}
        const char *what() const throw() { return msg.c_str(); }

    private:
        std::string msg;
    };

    /// reference typedefs.h:323-375
    class ConstraintInfo
    {
    public:
        ConstraintInfo() : obj_index(0), ctr_index(0) {}
        ConstraintInfo(int o, int c) : obj_index(o), ctr_index(c) {}
        void increment_obj_index(int increment) { obj_index += increment; }
        void set_ctr_index(int c) { ctr_index = c; }
        int get_obj_index() const { return obj_index; }
        int get_ctr_index() const { return ctr_index; }
        friend bool operator==(const ConstraintInfo &a, const ConstraintInfo &b)
        {
            return a.obj_index == b.obj_index && a.ctr_index == b.ctr_index;
        }

    private:
        int obj_index;
        int ctr_index;
    };

    /// reference typedefs.h:380-432
    class WorkingSetLogEntry
    {
    public:
        WorkingSetLogEntry() {}
        WorkingSetLogEntry(Index o, Index c, ConstraintActivationType t, RealScalar a, Index r)
        : obj_index(o), ctr_index(c), ctr_type(t), alpha_or_lambda(a), rank(r), cycling_detected(false)
        {
        }
        Index obj_index;
        Index ctr_index;
        ConstraintActivationType ctr_type;
        RealScalar alpha_or_lambda;
        Index rank;
        bool cycling_detected;
    };

    /// reference typedefs.h:439-527 (comparison ignores alpha_or_lambda, :470-488)
    class ConstraintIdentifier
    {
    public:
        ConstraintIdentifier() : obj_index(0), ctr_index(0), ctr_type(CTR_INACTIVE), alpha_or_lambda(0), cycling_detected(false) {}
        ConstraintIdentifier(Index o, Index c, ConstraintActivationType t, RealScalar a = 0)
        : obj_index(o), ctr_index(c), ctr_type(t), alpha_or_lambda(a), cycling_detected(false)
        {
        }
        void set(Index o, Index c, ConstraintActivationType t)
        {
            obj_index = o;
            ctr_index = c;
            ctr_type  = t;
        }
        bool operator==(const ConstraintIdentifier &ci) const
        {
            return obj_index == ci.obj_index && ctr_index == ci.ctr_index && ctr_type == ci.ctr_type;
        }
        Index obj_index;
        Index ctr_index;
        ConstraintActivationType ctr_type;
        RealScalar alpha_or_lambda;
        bool cycling_detected;
    };

    namespace internal
    {
        enum OperationType
        {
            OPERATION_UNDEFINED,
            OPERATION_ADD,
            OPERATION_REMOVE
        };

        /// reference typedefs.h:621-670
        class ObjectiveInfo
        {
        public:
            ObjectiveInfo() : dim(0), rank(0), first_row_index(0), first_col_index(0), regularization_factor(0.0) {}
            Index dim;
            Index rank;
            Index first_row_index;
            Index first_col_index;
            RealScalar regularization_factor;
        };

        /// reference utility.h:48-51
        inline bool isEqual(RealScalar a, RealScalar b, RealScalar tol = 1e-15) { return std::abs(a - b) < tol; }
    } // namespace internal
} // namespace LexLS
