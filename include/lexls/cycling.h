/*
 * Derived work: this header restates, for a different equality-solver back end, host-side interface and control flow of
 * jrl-umi3218/lexls (include/lexls/cycling.h), whose notice is retained as its BSD 3-clause licence requires:
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
// Cycling detection for the active-set driver (host side; off by default, typedefs.h:277).
//
// Behaviour of the reference's include/lexls/cycling.h:32-65, restated: the driver reports every working-set change; when a
// constraint is ADDED right after the very same (objective, constraint, type) was REMOVED, the driver is going round in a
// circle.  The answer is to relax the bound that was just re-activated by `relax_step` (Objective::relax_bounds) and to count
// the event; after `max_counter` relaxations the solve ends with PROBLEM_SOLVED_CYCLING_HANDLING.
#pragma once

#include <lexls/objective.h>

namespace LexLS
{
    namespace internal
    {
        class CyclingHandler
        {
        public:
            CyclingHandler() : relaxations_done(0), relaxations_allowed(50), step(1e-08) { forget(); }

            void reset()
            {
                relaxations_done = 0;
                forget();
            }

            /// one working-set change; `cycling_detected` tells whether it closed a REMOVE -> ADD circle (and a bound was relaxed)
            TerminationStatus update(OperationType operation, ConstraintIdentifier what, std::vector<Objective> &objectives, bool &cycling_detected)
            {
                const bool circle = last.valid && operation == OPERATION_ADD && last.operation == OPERATION_REMOVE && what == last.what;
                cycling_detected  = false;
                if (circle && relaxations_done >= relaxations_allowed) return PROBLEM_SOLVED_CYCLING_HANDLING; // (the event is not recorded)
                if (circle)
                {
                    objectives[last.what.obj_index].relax_bounds(last.what.ctr_index, last.what.ctr_type, step);
                    relaxations_done++;
                    cycling_detected = true;
                }
                last.valid     = true;
                last.operation = operation;
                last.what      = what;
                return TERMINATION_STATUS_UNKNOWN;
            }

            void set_max_counter(Index m) { relaxations_allowed = m; }
            void set_relax_step(RealScalar s) { step = s; }
            Index get_counter() const { return relaxations_done; }

        private:
            struct Event
            {
                bool valid;
                OperationType operation;
                ConstraintIdentifier what;
            };
            void forget()
            {
                last.valid     = false;
                last.operation = OPERATION_UNDEFINED;
                last.what.set(0, 0, CTR_INACTIVE);
            }

            Index relaxations_done, relaxations_allowed;
            RealScalar step;
            Event last;
        };
    } // namespace internal
} // namespace LexLS
