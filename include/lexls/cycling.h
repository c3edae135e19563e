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
