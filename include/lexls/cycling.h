// Cycling detection for the active-set driver (host side; off by default, typedefs.h:277).
// Behaviour restated from the reference include/lexls/cycling.h:32-65: an ADD that immediately
// follows a REMOVE of the very same (objective, constraint, type) is a cycle; the bound that was
// just re-activated is relaxed by relax_step, at most max_counter times.
#pragma once

#include <lexls/objective.h>

namespace LexLS
{
    namespace internal
    {
        class CyclingHandler
        {
        public:
            CyclingHandler() : counter(0), max_counter(50), relax_step(1e-08), previous_operation(OPERATION_UNDEFINED)
            {
                previous_ctr_identifier.set(0, 0, CTR_INACTIVE);
            }

            void reset()
            {
                counter            = 0;
                previous_operation = OPERATION_UNDEFINED;
                previous_ctr_identifier.set(0, 0, CTR_INACTIVE);
            }

            TerminationStatus update(OperationType operation, ConstraintIdentifier ctr_identifier, std::vector<Objective> &Obj, bool &cycling_detected)
            {
                cycling_detected = false;
                if (operation == OPERATION_ADD && previous_operation == OPERATION_REMOVE && ctr_identifier == previous_ctr_identifier)
                {
                    if (counter >= max_counter) return PROBLEM_SOLVED_CYCLING_HANDLING;
                    Obj[previous_ctr_identifier.obj_index].relax_bounds(previous_ctr_identifier.ctr_index, previous_ctr_identifier.ctr_type, relax_step);
                    counter++;
                    cycling_detected = true;
                }
                previous_operation      = operation;
                previous_ctr_identifier = ctr_identifier;
                return TERMINATION_STATUS_UNKNOWN;
            }

            void set_max_counter(Index m) { max_counter = m; }
            void set_relax_step(RealScalar s) { relax_step = s; }
            Index get_counter() const { return counter; }

        private:
            Index counter;
            Index max_counter;
            RealScalar relax_step;
            OperationType previous_operation;
            ConstraintIdentifier previous_ctr_identifier;
        };
    } // namespace internal
} // namespace LexLS
