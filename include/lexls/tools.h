/*
 * Derived work: this header restates, for a different equality-solver back end, host-side interface and control flow of
 * jrl-umi3218/lexls (include/lexls/tools.h), whose notice is retained as its BSD 3-clause licence requires:
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
// Reader for the reference's ASCII hierarchy format (".dat"), host side.
//
// Format as consumed by the reference's tools::HierarchyFileProcessor::import (tools.h:261-453):
//   header fields, any order, value on the next line:  #nVar  #nObj  #nCtr (list)  #HierType
//   {100 equalities | 200 inequalities | 210 inequalities + active-set guess}  #ObjType (list of
//   100 simple bounds | 200 general);  then "#OBJECTIVE k" blocks in ascending order, one row per
//   line: equalities [a_1..a_n b], inequalities [a_1..a_n lb ub], simple bounds [var lb ub]
//   (only objective 0), with a trailing activation flag 0..3 per row for type 210; then optional
//   "#SolGuess" and "#Solution" blocks of nVar numbers.
// Excess columns are ignored, missing ones are an error (tools.h:196-216).
//
// Note: simple-bound variable indices are stored as written in the file; the
// reference's own test file uses 1-based indices, which the caller converts (see to_zero_based).
#pragma once

#include <fstream>
#include <lexls/typedefs.h>
#include <sstream>
#include <stdexcept>

namespace LexLS
{
    namespace tools
    {
        enum HierarchyType
        {
            HIERARCHY_TYPE_NONE       = 0,
            HIERARCHY_TYPE_EQUALITY   = 1,
            HIERARCHY_TYPE_INEQUALITY = 2
        };

        struct Hierarchy
        {
            HierarchyType type_of_hierarchy;
            unsigned int type_header; // 100 / 200 / 210
            Index number_of_variables;
            Index number_of_objectives;
            std::vector<Index> number_of_constraints;
            std::vector<ObjectiveType> types_of_objectives;
            std::vector<dMatrixType> objectives;
            std::vector<std::vector<ConstraintActivationType>> active_set_guess;
            dVectorType solution_guess;
            dVectorType solution;
        };

        class HierarchyFileProcessor
        {
        public:
            void import(const std::string &file_name, Hierarchy &h) const
            {
                std::ifstream ifs(file_name.c_str());
                if (!ifs) throw std::runtime_error("Cannot open file for reading");

                h                      = Hierarchy();
                h.type_of_hierarchy    = HIERARCHY_TYPE_NONE;
                h.type_header          = 0;
                h.number_of_variables  = 0;
                h.number_of_objectives = 0;

                bool got_nvar = false, got_nobj = false, got_nctr = false, got_type = false, got_objtype = false;
                std::string line;

                while (!(got_nvar && got_nobj && got_nctr && got_type && got_objtype) && std::getline(ifs, line))
                {
                    const std::string key = strip(line);
                    if (key == "#nVar")
                    {
                        once(got_nvar);
                        h.number_of_variables = static_cast<Index>(read_uints(ifs, 1)[0]);
                    }
                    else if (key == "#nObj")
                    {
                        once(got_nobj);
                        h.number_of_objectives = static_cast<Index>(read_uints(ifs, 1)[0]);
                    }
                    else if (key == "#HierType")
                    {
                        once(got_type);
                        h.type_header = read_uints(ifs, 1)[0];
                        if (h.type_header == 100)
                            h.type_of_hierarchy = HIERARCHY_TYPE_EQUALITY;
                        else if (h.type_header == 200 || h.type_header == 210)
                            h.type_of_hierarchy = HIERARCHY_TYPE_INEQUALITY;
                        else
                            throw std::runtime_error("Unsupported type of hierarchy.");
                    }
                    else if (key == "#nCtr")
                    {
                        once(got_nctr);
                        const std::vector<unsigned int> v = read_uints(ifs, 0);
                        h.number_of_constraints.assign(v.begin(), v.end());
                    }
                    else if (key == "#ObjType")
                    {
                        once(got_objtype);
                        const std::vector<unsigned int> v = read_uints(ifs, 0);
                        for (size_t k = 0; k < v.size(); k++)
                        {
                            if (v[k] == 100)
                                h.types_of_objectives.push_back(SIMPLE_BOUNDS_OBJECTIVE);
                            else if (v[k] == 200)
                                h.types_of_objectives.push_back(GENERAL_OBJECTIVE);
                            else
                                throw std::runtime_error("Unsupported type of objective.");
                        }
                    }
                }
                if (!(got_nvar && got_nobj && got_nctr && got_type && got_objtype)) throw std::runtime_error("At least one required parameters is not set.");
                if (h.types_of_objectives.size() != h.number_of_objectives || h.number_of_constraints.size() != h.number_of_objectives)
                    throw std::runtime_error("Wrong number of objectives.");

                const unsigned int number_of_bounds = (h.type_header == 100) ? 1 : 2;
                h.objectives.resize(h.number_of_objectives);
                if (h.type_header == 210) h.active_set_guess.resize(h.number_of_objectives);

                Index k = 0;
                while (k < h.number_of_objectives && std::getline(ifs, line))
                {
                    if (strip(line).compare(0, 10, "#OBJECTIVE") != 0) continue;
                    Index ncols;
                    if (h.types_of_objectives[k] == SIMPLE_BOUNDS_OBJECTIVE)
                    {
                        if (k != 0) throw std::runtime_error("Simple constraints are supported only in the first objective.");
                        ncols = 1 + number_of_bounds;
                    }
                    else
                    {
                        ncols = h.number_of_variables + number_of_bounds;
                    }
                    const Index nrows = h.number_of_constraints[k];
                    h.objectives[k].resize(nrows, ncols);
                    if (h.type_header == 210) h.active_set_guess[k].assign(nrows, CTR_INACTIVE);
                    for (Index r = 0; r < nrows; r++)
                    {
                        if (!std::getline(ifs, line)) throw std::runtime_error("Not enough data.");
                        std::istringstream ls(line);
                        for (Index c = 0; c < ncols; c++)
                        {
                            double val;
                            if (!(ls >> val)) throw std::runtime_error("Not enough data.");
                            h.objectives[k](r, c) = val;
                        }
                        if (h.type_header == 210)
                        {
                            unsigned int flag;
                            if (ls >> flag)
                            {
                                if (flag > 3) throw std::runtime_error("Unsupported constraint activation type.");
                                h.active_set_guess[k][r] = static_cast<ConstraintActivationType>(flag);
                            }
                        }
                    }
                    k++;
                }
                if (k != h.number_of_objectives) throw std::runtime_error("The number of objectives is lower than expected.");

                while (std::getline(ifs, line))
                {
                    const std::string key = strip(line);
                    if (key == "#SolGuess")
                        read_vector(ifs, h.number_of_variables, h.solution_guess);
                    else if (key == "#Solution")
                        read_vector(ifs, h.number_of_variables, h.solution);
                }
            }

            /// simple-bound indices of objective 0 as 0-based integers; `one_based` says how the file
            /// stores them (the reference's MEX subtracts 1, interfaces/matlab-octave/lexlsi.cpp:412)
            static std::vector<Index> simple_bound_indices(const Hierarchy &h, bool one_based)
            {
                std::vector<Index> idx;
                if (h.number_of_objectives == 0 || h.types_of_objectives[0] != SIMPLE_BOUNDS_OBJECTIVE) return idx;
                for (Index r = 0; r < h.objectives[0].rows(); r++)
                    idx.push_back(static_cast<Index>(std::llround(h.objectives[0](r, 0))) - (one_based ? 1u : 0u));
                return idx;
            }

        private:
            static std::string strip(const std::string &s)
            {
                std::string out;
                for (size_t i = 0; i < s.size(); i++)
                    if (!isspace(static_cast<unsigned char>(s[i]))) out.push_back(s[i]);
                return out;
            }
            static void once(bool &flag)
            {
                if (flag) throw std::runtime_error("Duplicate header field.");
                flag = true;
            }
            /// reads one line of unsigned integers; count == 0 means "as many as there are"
            static std::vector<unsigned int> read_uints(std::ifstream &ifs, size_t count)
            {
                std::string line;
                std::vector<unsigned int> out;
                while (out.empty() && std::getline(ifs, line))
                {
                    std::istringstream ls(line);
                    unsigned int v;
                    while (ls >> v) out.push_back(v);
                }
                if (out.empty() || (count && out.size() < count)) throw std::runtime_error("Could not read a header value.");
                return out;
            }
            static void read_vector(std::ifstream &ifs, Index n, dVectorType &v)
            {
                v.resize(n);
                for (Index i = 0; i < n; i++)
                    if (!(ifs >> v(i))) throw std::runtime_error("Could not read a solution vector.");
            }
        };
    } // namespace tools
} // namespace LexLS
