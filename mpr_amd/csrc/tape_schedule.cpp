/* tape_schedule.cpp — see tape_schedule.hpp */
#include "tape_schedule.hpp"

#include <algorithm>
#include <numeric>

#include "../../include/mpr_clause.h"

namespace mpr {

TapeSchedule build_schedule(const uint64_t* clauses, int32_t length)
{
    TapeSchedule s;
    const int32_t n = length - 2;
    if (n < 1 || n + 3 > 65535) return s;
    int32_t lastdef[256];
    for (int i = 0; i < 256; ++i) lastdef[i] = 0;      /* a slot read before it is written: X */
    const uint64_t head = clauses[0];
    lastdef[mpr_cl_out(head)] = 0;
    lastdef[mpr_cl_lhs(head)] = 1;
    lastdef[mpr_cl_rhs(head)] = 2;
    std::vector<int32_t> level((size_t)n + 3, 0);
    std::vector<SchedRec> recs((size_t)n);
    int32_t ord = 0, depth = 0;
    s.prev_writer.assign((size_t)n, 0xFFFF);
    s.defs.assign((size_t)n, 0);
    for (int32_t i = 0; i < n; ++i) {
        const uint64_t c = clauses[1 + i];
        const uint32_t op = mpr_cl_op(c), o = mpr_cl_out(c), l = mpr_cl_lhs(c), r = mpr_cl_rhs(c);
        if (op < 2 || op >= MPR_OP_COUNT) return s;    /* jump / terminator / unknown inside the body */
        SchedRec& q = recs[(size_t)i];
        q.clause = c;
        q.pl = (uint16_t)(l ? lastdef[l] : 0);
        q.pr = (uint16_t)(r ? lastdef[r] : 0);
        q.idx = (uint16_t)i;
        q.ord = 0;
        if (mpr_op_is_minmax(op)) {
            if (ord > 65535) return s;
            q.ord = (uint16_t)ord++;
        }
        int32_t lv = 0;
        if (l) lv = std::max(lv, level[q.pl]);
        if (r) lv = std::max(lv, level[q.pr]);
        level[(size_t)3 + i] = lv + 1;
        depth = std::max(depth, lv + 1);
        s.defs[(size_t)i] = (uint32_t)q.pl | ((uint32_t)q.pr << 16);
        if (lastdef[o] >= 3) s.prev_writer[(size_t)i] = (uint16_t)(lastdef[o] - 3);
        lastdef[o] = 3 + i;
    }
    s.root_val = lastdef[mpr_cl_out(clauses[length - 1])];
    /* stable sort by (level, opcode): lanes of one wave mostly run the same operation */
    std::vector<int32_t> order((size_t)n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        const int32_t la = level[(size_t)3 + a], lb = level[(size_t)3 + b];
        if (la != lb) return la < lb;
        return mpr_cl_op(recs[(size_t)a].clause) < mpr_cl_op(recs[(size_t)b].clause);
    });
    s.recs.resize((size_t)n);
    s.level_start.assign((size_t)depth + 1, 0);
    for (int32_t k = 0; k < n; ++k) {
        s.recs[(size_t)k] = recs[(size_t)order[(size_t)k]];
        s.level_start[(size_t)level[(size_t)3 + order[(size_t)k]]]++;    /* counts at [level]; level >= 1 */
    }
    /* counts -> offsets: level_start[L-1] = first record of level L */
    int32_t acc = 0;
    for (int32_t L = 1; L <= depth; ++L) {
        const int32_t cnt = s.level_start[(size_t)L];
        s.level_start[(size_t)L - 1] = acc;
        acc += cnt;
    }
    s.level_start[(size_t)depth] = acc;
    s.nclauses = n;
    s.ok = true;
    return s;
}

}  // namespace mpr
