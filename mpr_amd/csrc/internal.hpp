/* internal.hpp — definitions shared by the translation units of libmpr_amd.so */
#pragma once
#include <atomic>
#include <cstdint>
#include <string>
#include <vector>

#include "tape_schedule.hpp"

struct mpr_tape {
    std::vector<uint64_t> clauses;   /* head, operations, end — host copy */
    int32_t num_slots = 0;           /* highest slot index used + 1 */
    int32_t num_choices = 0;         /* min/max clauses */
    int32_t flags = 0;
    uint64_t serial = 0;             /* identity, so a context can cache per-tape state */
    mpr::TapeSchedule schedule;      /* dependency levels for the wide first-stage kernel (ok == false: not usable) */
};

namespace mpr {
int set_error(int code, const std::string& msg);
}
