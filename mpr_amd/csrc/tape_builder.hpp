/* tape_builder.hpp — DAG -> clause tape (host).  See tape_builder.cpp. */
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "tree.hpp"

namespace mpr {
namespace front {

struct TapeBuild {
    std::vector<uint64_t> clauses;   /* head clause, operations, end clause */
    int num_slots = 0;               /* slots used, including reserved slot 0 */
    bool slots_exhausted = false;    /* src/tape.cpp:79-81 "Ran out of slots!" */
    int unsupported = 0;             /* opcodes src/tape.cpp:182-196 does not implement */
    std::string warnings;
    std::string error;
};

TapeBuild build_tape(const Tree& tree);

}  // namespace front
}  // namespace mpr
