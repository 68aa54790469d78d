/* tape_schedule.hpp — dependency levels of a root tape, for the wide stage-0 kernel
 * (kernels_wide.hip).
 *
 * A tape is a straight-line program over a small slot file (src/tape.cpp:21-228 allocates the
 * slots).  Renaming every clause's operands to the CLAUSE that produced them ("value" indices:
 * 0, 1, 2 = X, Y, Z; 3 + i = result of clause i) removes the slot reuse, and what is left is a
 * DAG whose depth is tiny next to its length (prospero: 6056 clauses, 22 levels).  Clauses of
 * one level are independent, so a whole workgroup can evaluate ONE tile level by level instead
 * of one lane walking the tape clause by clause.  Nothing about the arithmetic changes: every
 * clause sees the operand values it would have seen in tape order.
 */
#pragma once
#include <cstdint>
#include <vector>

namespace mpr {

struct SchedRec {          /* 16 bytes, one per clause, stored in (level, opcode) order */
    uint64_t clause;
    uint16_t pl, pr;       /* value index of lhs / rhs (meaningless when the slot byte is 0) */
    uint16_t idx;          /* index of the clause in the tape body */
    uint16_t ord;          /* min/max clauses: ordinal among the tape's min/max clauses (choice index) */
};

struct TapeSchedule {
    bool ok = false;                    /* false: tape not schedulable this way (too long, jumps, ...) */
    int32_t nclauses = 0;               /* body length */
    int32_t root_val = 0;               /* value index of the result */
    std::vector<SchedRec> recs;
    std::vector<int32_t> level_start;   /* nlevels + 1 offsets into recs */
    std::vector<uint16_t> prev_writer;  /* per clause (tape order): the previous clause with the same out slot, 0xFFFF = none */
    std::vector<uint32_t> defs;         /* per clause (tape order): pl | pr << 16, as in its record */
};

/* clauses = head, body..., end */
TapeSchedule build_schedule(const uint64_t* clauses, int32_t length);

}  // namespace mpr
