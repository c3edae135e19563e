/*
 * Derived work: this header restates, for a different equality-solver back end, host-side interface and control flow of
 * jrl-umi3218/lexls (include/lexls/workingset.h), whose notice is retained as its BSD 3-clause licence requires:
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
// Working set of one LexLSI objective (host side, kept on the host by design).
//
// Behavioural contract restated from the reference include/lexls/workingset.h:
//  * activate(): the constraint leaves `inactive` by swap-with-last (order of the remaining
//    inactive constraints changes) and is appended to `active`            (workingset.h:57-78)
//  * deactivate(): order-preserving erase from `active`, append to `inactive` (workingset.h:91-115)
//  * getCtrIndex(): linear search in whichever list holds the constraint   (workingset.h:154-168)
// These ordering rules decide which blocking constraint / multiplier wins a tie, hence active-set
// index parity.
#pragma once

#include <lexls/typedefs.h>
#include <numeric>

namespace LexLS
{
    namespace internal
    {
        class WorkingSet
        {
        public:
            void resize(Index dim)
            {
                type_of.assign(dim, CTR_INACTIVE);
                act.reserve(dim);
                act_type.reserve(dim);
                reset();
            }

            void reset()
            {
                std::fill(type_of.begin(), type_of.end(), CTR_INACTIVE);
                act.clear();
                act_type.clear();
                inact.resize(type_of.size());
                std::iota(inact.begin(), inact.end(), 0u);
            }

            void activate(Index CtrIndex, ConstraintActivationType type)
            {
                if (type_of[CtrIndex] != CTR_INACTIVE) throw Exception("Cannot activate an active constraint");
                const Index pos = getCtrIndex(CtrIndex);
                inact[pos]      = inact.back();
                inact.pop_back();
                type_of[CtrIndex] = type;
                act.push_back(CtrIndex);
                act_type.push_back(type);
            }

            void deactivate(Index CtrIndexActive)
            {
                const Index CtrIndex = act[CtrIndexActive];
                if (type_of[CtrIndex] == CTR_INACTIVE) throw Exception("Cannot deactivate an inactive constraint");
                act.erase(act.begin() + CtrIndexActive);
                act_type.erase(act_type.begin() + CtrIndexActive);
                type_of[CtrIndex] = CTR_INACTIVE;
                inact.push_back(CtrIndex);
            }

            Index getActiveCtrCount() const { return static_cast<Index>(act.size()); }
            Index getActiveCtrIndex(Index k) const { return act[k]; }
            ConstraintActivationType getActiveCtrType(Index k) const { return act_type[k]; }
            ConstraintActivationType getCtrType(Index k) const { return type_of[k]; }
            Index getInactiveCtrCount() const { return static_cast<Index>(inact.size()); }
            Index getInactiveCtrIndex(Index k) const { return inact[k]; }
            bool isActive(Index k) const { return type_of[k] != CTR_INACTIVE; }

            /// position of constraint k inside the list (active or inactive) that currently holds it
            Index getCtrIndex(Index k) const
            {
                const std::vector<Index> &list = isActive(k) ? act : inact;
                return static_cast<Index>(std::find(list.begin(), list.end(), k) - list.begin());
            }

        private:
            std::vector<Index> act;
            std::vector<Index> inact;
            std::vector<ConstraintActivationType> act_type;
            std::vector<ConstraintActivationType> type_of;
        };
    } // namespace internal
} // namespace LexLS
