/*
 * kernels_wide.hip — eval_tiles_i for the FIRST tile stage, one workgroup per tile.
 *
 * The reference evaluates a tile with one thread that walks the tape clause by clause
 * (src/context.cu:77-458), and so does k_eval_tiles (kernels.hip), one lane per tile.  At the
 * first stage there are only (S/64)^2 or (S/64)^3 tiles — 256 at 1024^2 — and every one walks the
 * whole root tape: four wavefronts stepping through prospero's 6056 clauses twice is 3.9 ms of
 * pure dependent-instruction latency on an otherwise idle chip, 90 % of that frame.
 *
 * The root tape's dependency DAG is shallow (tape_schedule.hpp: prospero 22 levels, bear 72), so
 * here the 256 threads of a workgroup evaluate one tile LEVEL BY LEVEL:
 *   forward   every thread takes clauses of the current level (16-byte records: clause, operand
 *             value indices, clause index, choice ordinal; records of a level are sorted by
 *             opcode), reads its operands' intervals from the value array in LDS, runs the same
 *             interval_clause() as everybody else (device_math.hpp, round-up mode), stores the
 *             result interval and the min/max choice;
 *   classify  empty / masked / filled / ambiguous exactly as reference :293-321;
 *   shorten   (ambiguous tiles that made a choice) liveness is propagated from the result down
 *             the levels — the same def-use reachability the reference's backward walk computes
 *             with its per-slot "active" flags (:351-458) — then the live clauses are ranked in
 *             reverse tape order by a block-wide prefix sum and written to freshly claimed chunks
 *             in the reference's layout: 62 clauses per 64-word chunk, links in word 0 and 63,
 *             decided min/max clauses rewritten to COPY_LHS / COPY_RHS / COPY_IMM or dropped when
 *             the copy would be onto itself.
 * The values, choices, tile states and shortened tapes are those of the serial walk; the tests
 * compare both kernels and the oracle (tests/test_gpu_render.py).
 *
 * LATER stages (while a stage has few tiles: small images, a rank's share of a multi-GPU frame).  A
 * shortened tape is a sub-sequence of the root tape with some min / max clauses turned into copies
 * (:351-458), so a tile of any stage can be evaluated on the ROOT tape's level schedule, given two bits
 * per root clause for the tape it inherited: 0 not in the tape, 1 in the tape as it is, 2 / 3 decided
 * for lhs / rhs (a COPY in the tape, or nothing when the copy would be onto itself).  Every ambiguous
 * tile writes that table for the tape it leaves to its children (bits_out, indexed by the tile) next
 * to the tape itself in the reference's layout, and its children (which find their parent's index in
 * the `next` field of their node until their own stage's compaction overwrites it) read it back
 * (bits_in).  Stage after stage runs like this while the previous stage did; the serial walk, the
 * float pass and the normals pass read the reference-layout tapes as before.
 */
#include "kernel_common.hpp"

namespace mprk {

/* one byte per clause in LDS: bits 0-1 this tile's own choice, bits 2-3 the inherited state (see the header),
 * bits 4-5 liveness (1 live, 2 live but not written).  Phases that change different fields are separated by
 * barriers; concurrent marks of one byte all set the same bit. */
DEV void mark_val(unsigned char* st, uint32_t v)
{
    if (v >= 3) st[v - 3] |= 0x10;
}
DEV int live_of(unsigned char b) { return b >> 4; }
/* a clause's decision, inherited (bits 2-3: 2 lhs, 3 rhs) or this tile's own (bits 0-1): 0 none, 1 lhs, 2 rhs */
DEV int total_choice(unsigned char b)
{
    const int st = (b >> 2) & 3;
    return st >= 2 ? st - 1 : (b & 3);
}

template <int DIM>
__global__ void __launch_bounds__(1024)
k_eval_tiles_wide(WideStageArgs w)
{
    const TileStageArgs& a = w.t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = w.nclauses;
    float2* const V = reinterpret_cast<float2*>(smem);                                   /* [n + 3] values */
    unsigned char* const ch = smem + (size_t)(n + 3) * 8;                                /* [n] state of clause i (mark_val) */
    int* const sh = reinterpret_cast<int*>(smem + (((size_t)(n + 3) * 8 + (size_t)n + 15) & ~(size_t)15));   /* [1040] */
    /* sh[0] any choice, sh[1] state (0 dead, 1 ambiguous), sh[2] pool base, sh[3] ok, sh[4] min live-and-dropped i,
     * sh[5] min written i, sh[8..8+1024] scan */

    const int tid = threadIdx.x;
    const int nt = blockDim.x;                              /* 64 for narrow tapes, 256 for wide ones, 1024 when the tiles are few */
    const int gidx = blockIdx.x;
    const uint64_t* __restrict__ const tro = a.tape_ro;
    uint64_t* __restrict__ const twr = a.tape_wr;

    /* tile state: read once (other workgroups update the image while this one runs) */
    if (tid == 0) {
        const mpr_tile_node node = a.tiles[gidx];
        int alive = node.position != -1;
        if (DIM == 3 && alive && !a.no_mask) {
            const int4_ p = unpack(node.position, a.tps);
            if (a.image[p.w] > p.z) {                       /* mask_filled_tiles before evaluation */
                alive = 0;
                a.tiles[gidx].position = -1;
            }
        }
        sh[0] = 0;
        sh[1] = alive;
        sh[6] = node.position;
        sh[7] = node.tape;
        sh[1037] = node.next;                               /* with bits_in: the parent's index in its stage's list */
        sh[4] = 0x7FFFFFFF;
        sh[5] = 0x7FFFFFFF;
        sh[1036] = 0;                                       /* min / max clauses the shortened tape keeps undecided */
    }
    __syncthreads();
    if (sh[1] == 0) return;
    const int4_ pos = unpack(sh[6], a.tps);
    const int tape = sh[7];                                 /* the tape the tile inherited (first stage: the root tape) */
    const int root = w.root_tape;
    /* ch[i]: bits 0-1 this tile's own choice at clause i, bits 2-3 the inherited state (see the header) */
    const uint32_t* __restrict__ const inherited = w.bits_in ? w.bits_in + (size_t)sh[1037] * w.wpt : nullptr;
    for (int i = tid; i < n; i += nt) {
        const uint32_t st = inherited ? (inherited[i >> 4] >> ((i & 15) * 2)) & 3u : 1u;
        ch[i] = (unsigned char)(st << 2);
    }

    /* tile corners in round-to-nearest (reference :91-96), then the view transform in round-up mode */
    const float t = (float)a.tps;
    float c0 = (pos.x / t - 0.5f) * 2.0f, c1 = ((pos.x + 1) / t - 0.5f) * 2.0f;
    float c2 = (pos.y / t - 0.5f) * 2.0f, c3 = ((pos.y + 1) / t - 0.5f) * 2.0f;
    float c4 = 0.0f, c5 = 0.0f;
    if (DIM == 3) {
        c4 = (pos.z / t - 0.5f) * 2.0f;
        c5 = ((pos.z + 1) / t - 0.5f) * 2.0f;
    }
    round_up_begin(c0, c1, c2, c3, c4, c5);
    /* ---- from here on: f32 round-up mode, rounded f32 arithmetic only through device_math ---- */
    if (tid == 0) {
        const ival ix = iv(c0, c1), iy = iv(c2, c3), iz = iv(c4, c5);
        ival vx, vy, vz;
        if (DIM == 3) {
            ival r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                r[i] = i_add_f(i_add(i_add(i_mul_f(ix, a.mat[i]), i_mul_f(iy, a.mat[i + 4])),
                                     i_mul_f(iz, a.mat[i + 8])), a.mat[i + 12]);
            }
            vx = i_div(r[0], r[3]);
            vy = i_div(r[1], r[3]);
            vz = i_div(r[2], r[3]);
        } else {
            ival r[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                r[i] = i_add_f(i_add(i_mul_f(ix, a.mat[i]), i_mul_f(iy, a.mat[i + 3])), a.mat[i + 6]);
            }
            vx = i_div(r[0], r[2]);
            vy = i_div(r[1], r[2]);
            vz = iv(a.z, a.z);
        }
        V[0] = make_float2(vx.lo, vx.hi);
        V[1] = make_float2(vy.lo, vy.hi);
        V[2] = make_float2(vz.lo, vz.hi);
    }
    __syncthreads();

    /* ---- forward, level by level ---- */
    const uint4* __restrict__ const recs = reinterpret_cast<const uint4*>(w.recs);
    bool my_choice = false;
    /* the first record of the NEXT level is fetched before this level's arithmetic: a level is one
     * clause per thread more often than not, and the fetch (an L2 hit, every tile reads the same
     * records) would otherwise sit on the critical path of every level */
    int begin = w.level_start[0], end = w.nlevels > 0 ? w.level_start[1] : 0;
    uint4 qn = make_uint4(0, 0, 0, 0);
    if (begin + tid < end) qn = recs[begin + tid];
    for (int lv = 0; lv < w.nlevels; ++lv) {
        const int nbegin = end, nend = (lv + 1 < w.nlevels) ? w.level_start[lv + 2] : end;
        uint4 q = qn;
        if (nbegin + tid < nend) qn = recs[nbegin + tid];
        for (int k = begin + tid; k < end; k += nt) {
            if (k != begin + tid) q = recs[k];
            const uint32_t op = q.x & 0xFF, r8 = q.x >> 24;
            const uint32_t pl = q.z & 0xFFFF, pr = q.z >> 16, idx = q.w & 0xFFFF, ord = q.w >> 16;
            const float imm = mpr_u2f(q.y);
            const uint32_t st = (ch[idx] >> 2) & 3;
            if (st == 0) continue;                                     /* not in this tile's tape */
            const float2 lv2 = V[pl];
            const float2 rv2 = V[pr];
            const ival B = r8 ? iv(rv2.x, rv2.y) : iv(imm, imm);       /* immediate forms carry rhs == 0 */
            if (st != 1) {                                             /* decided higher up: COPY_LHS / COPY_RHS / COPY_IMM */
                V[3 + idx] = (st == 2) ? lv2 : make_float2(B.lo, B.hi);
                continue;
            }
            int c = 0;
            const ival out = interval_clause(op, iv(lv2.x, lv2.y), B, imm, c);
            V[3 + idx] = make_float2(out.lo, out.hi);
            /* choices past choice_cap are not recorded by the reference (:257): they still make the
             * tile push a tape, but the backward pass keeps both sides of such a clause.  (A shortened
             * tape has no more min / max clauses than the stage's cap: context.hip, stage_choice_cap.) */
            ch[idx] = (unsigned char)((1u << 2) | ((inherited || (int)ord < a.choice_cap) ? c : 0));
            my_choice |= c != 0;
        }
        __syncthreads();
        begin = nbegin;
        end = nend;
    }
    if (my_choice) sh[0] = 1;
    const float2 res = V[w.root_val];

    /* ---- classification (reference :293-321) ---- */
    if (tid == 0) {
        int state = 0;
        if (res.x > 0.0f) {                                   /* empty */
            a.tiles[gidx].position = -1;
        } else if (DIM == 3 && !a.no_mask && __hip_atomic_load(&a.image[pos.w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > pos.z) {
            a.tiles[gidx].position = -1;                      /* masked */
        } else if (res.y < 0.0f) {                            /* filled */
            a.tiles[gidx].position = -1;
            if (DIM == 3) atomicMax(&a.image[pos.w], pos.z);
            else a.image[pos.w] = 1;
        } else {
            state = 1;
        }
        sh[1] = state;
        if (a.counters && !inherited) {
            atomicAdd((unsigned long long*)&a.counters[CNT_LANE], (unsigned long long)n);
            if ((gidx & 63) == 0) atomicAdd((unsigned long long*)&a.counters[CNT_FWD], (unsigned long long)(n + 1));
        }
    }
    if (a.counters && inherited) {                 /* the clauses of the inherited tape (jumps and end not counted) */
        int mine_in = 0;
        for (int i = tid; i < n; i += nt) mine_in += ((ch[i] >> 2) & 3) != 0;
        if (mine_in) {
            atomicAdd((unsigned long long*)&a.counters[CNT_LANE], (unsigned long long)mine_in);
            if ((gidx & 63) == 0) atomicAdd((unsigned long long*)&a.counters[CNT_FWD], (unsigned long long)mine_in);
        }
    }
    __syncthreads();
    uint32_t* __restrict__ const leave = w.bits_out ? w.bits_out + (size_t)gidx * w.wpt : nullptr;
    if (!(sh[1] == 1 && sh[0] != 0)) {             /* only ambiguous tiles that chose a side shorten their tape */
        /* an ambiguous tile that keeps its tape: the next stage needs room for all of its choices */
        if (sh[1] == 1 && tid == 0 && a.next_choices && a.choice_cap > 0 &&
            a.choice_cap > __hip_atomic_load(a.next_choices, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))       /* (kernels.hip: one word, many writers) */
            atomicMax(a.next_choices, a.choice_cap);
        if (sh[1] == 1 && leave) {                 /* ... and its children inherit what it inherited */
            for (int k = tid; k < w.wpt; k += nt) leave[k] = inherited ? inherited[k] : 0x55555555u;
        }
        return;
    }

    /* ---- liveness (reference :351-458, def-use form): breadth first from the result ----
     * The reference's walk keeps a flag per SLOT and marks both operand fields of every clause that is not a min / max
     * with a recorded choice — also of the COPY clauses an earlier stage left, whose unchosen field still names a slot.
     * Whatever clause of the inherited tape wrote that slot last is kept with them, related or not.  Same here: for such
     * a field the schedule's chain of earlier writers of the slot is followed to the first one the inherited tape holds.
     * The queue of clauses to visit lives where the values were (every clause enters it once). */
    uint16_t* const queue = reinterpret_cast<uint16_t*>(V);
    uint32_t* const chw = reinterpret_cast<uint32_t*>(ch);          /* ch is 8-byte aligned: live bits are set with word atomics */
    if (tid == 0) {
        sh[1038] = 0;
        if (w.root_val >= 3) {
            mark_val(ch, (uint32_t)w.root_val);
            queue[0] = (uint16_t)(w.root_val - 3);
            sh[1038] = 1;
        }
    }
    auto visit = [&](uint32_t v) {                                  /* value index of an operand */
        if (v < 3) return;
        const uint32_t j = v - 3;
        const uint32_t bit = 0x10u << (8 * (j & 3));
        if (atomicOr(&chw[j >> 2], bit) & bit) return;
        queue[atomicAdd(&sh[1038], 1)] = (uint16_t)j;
    };
    for (int head = 0;;) {
        __syncthreads();
        const int tail = sh[1038];
        __syncthreads();
        if (head >= tail) break;
        for (int e = head + tid; e < tail; e += nt) {
            const uint32_t i = queue[e];
            const uint32_t d = (uint32_t)tro[root + 1 + i], pq = w.defs[i];
            const uint32_t op = d & 0xFF, o8 = (d >> 8) & 0xFF, l8 = (d >> 16) & 0xFF, r8 = d >> 24;
            const uint32_t pl = pq & 0xFFFF, pr = pq >> 16;
            const uint32_t b = ch[i], st = (b >> 2) & 3;
            if (st >= 2) {                                            /* decided by an earlier stage */
                if (st == 2) visit(pl);
                else if (r8) visit(pr);
                const bool absent = (st == 2 && l8 == o8) || (st == 3 && r8 != 0 && r8 == o8);   /* a copy onto itself: never written */
                const uint32_t field = (st == 2) ? r8 : l8;       /* the slot the COPY clause still names */
                if (!absent && field) {
                    const uint32_t v = (st == 2) ? pr : pl;       /* the clause that wrote it in the root tape ... */
                    int j = v >= 3 ? (int)v - 3 : -1;
                    while (j >= 0) {                              /* ... or the last one before it that the inherited tape holds */
                        const uint32_t sj = (ch[j] >> 2) & 3;
                        bool held = sj == 1;
                        if (sj >= 2) {
                            const uint32_t dj = (uint32_t)tro[root + 1 + j];
                            const uint32_t oj = (dj >> 8) & 0xFF, lj = (dj >> 16) & 0xFF, rj = dj >> 24;
                            held = !((sj == 2 && lj == oj) || (sj == 3 && rj != 0 && rj == oj));
                        }
                        if (held) break;
                        const uint32_t pj = w.prev_writer[j];
                        j = pj == 0xFFFF ? -1 : (int)pj;
                    }
                    if (j >= 0) visit((uint32_t)j + 3);
                }
                continue;
            }
            const int c = mpr_op_is_minmax(op) ? (int)(b & 3) : 0;
            if (c == 1) {
                visit(pl);
            } else if (c == 2) {
                if (r8) visit(pr);
            } else {
                if (l8) visit(pl);
                if (r8) visit(pr);
            }
        }
        head = tail;
    }

    /* ---- rank the clauses that get written, in reverse tape order ---- */
    const int per = (n + nt - 1) / nt;
    const int hi_i = n - 1 - tid * per;                  /* this thread: i = hi_i, hi_i - 1, ... > lo_i */
    const int lo_i = max(hi_i - per, -1);
    int mine = 0, kept = 0;
    int min_drop = 0x7FFFFFFF, min_emit = 0x7FFFFFFF;
    for (int i = hi_i; i > lo_i; --i) {
        if (!live_of(ch[i])) continue;
        const uint64_t d = tro[root + 1 + i];
        const uint32_t op = (uint32_t)d & 0xFF, o = (uint32_t)(d >> 8) & 0xFF, l = (uint32_t)(d >> 16) & 0xFF,
                       r = (uint32_t)(d >> 24) & 0xFF;
        const int c = mpr_op_is_minmax(op) ? total_choice(ch[i]) : 0;
        kept += (mpr_op_is_minmax(op) && c == 0) ? 1 : 0;
        const bool drop = (c == 1 && l == o) || (c == 2 && r != 0 && r == o);
        if (drop) {
            min_drop = i;
            ch[i] = (unsigned char)((ch[i] & 0x0F) | 0x20);       /* live, nothing written */
        } else {
            ++mine;
            min_emit = i;
        }
    }
    sh[8 + tid] = mine;
    if (kept) atomicAdd(&sh[1036], kept);
    if (min_drop != 0x7FFFFFFF) atomicMin(&sh[4], min_drop);
    if (min_emit != 0x7FFFFFFF) atomicMin(&sh[5], min_emit);
    __syncthreads();
    /* exclusive scan of the per-thread counts (Hillis-Steele in LDS) */
    int incl = mine;
    for (int off = 1; off < nt; off <<= 1) {
        const int other = tid >= off ? sh[8 + tid - off] : 0;
        __syncthreads();
        incl += other;
        sh[8 + tid] = incl;
        __syncthreads();
    }
    const int total = sh[8 + nt - 1];
    int rank = incl - mine;                               /* clauses written before this thread's first */

    /* chunks: the end clause and 62 clauses in the first, 62 per further chunk; the reference opens
     * a chunk when a live clause arrives at a full one — even if that clause is then dropped */
    const bool spurious = total > 0 && total % 62 == 0 && sh[4] < sh[5];
    const int nchunks = (total == 0 ? 1 : (total - 1) / 62 + 1) + (spurious ? 1 : 0);
    if (tid == 0) {
        const unsigned long long want = (unsigned long long)MPR_SUBTAPE_CHUNK * nchunks;
        const unsigned long long base64 = atomicAdd(a.tape_index, want);      /* 64-bit index: see k_eval_tiles */
        int ok = base64 < (unsigned long long)a.pool_cap && base64 + want < 0x7FFFFFFFull;
        if (!ok) atomicAdd(a.tape_index, 0ull - want);       /* claims beyond the pool are handed back */
        else if (base64 + want >= (unsigned long long)a.pool_cap) ok = 0;
        const int base = ok ? (int)base64 : 0;
        sh[2] = base;
        sh[3] = ok;
        if (!ok) a.tape_index[1] = 1;
        if (!ok && a.counters) a.counters[CNT_OVERFLOW] = 1;
        if (a.next_choices) {
            const int need = ok ? sh[1036] : a.choice_cap;      /* pool exhausted: the tile keeps the root tape */
            if (need > 0 && need > __hip_atomic_load(a.next_choices, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.next_choices, need);
        }
    }
    __syncthreads();
    if (leave) {
        /* what the children inherit: the pushed tape's table, or (pool exhausted) the parent's again */
        for (int k = tid; k < w.wpt; k += nt) {
            uint32_t word = 0;
            if (!sh[3]) {
                word = inherited ? inherited[k] : 0x55555555u;
            } else {
                for (int j = 0; j < 16 && k * 16 + j < n; ++j) {
                    const int i = k * 16 + j;
                    const int c = total_choice(ch[i]);
                    word |= (uint32_t)(live_of(ch[i]) ? 1 + c : 0) << (2 * j);
                }
            }
            leave[k] = word;
        }
    }
    if (!sh[3]) return;                                   /* pool exhausted: the tile keeps its parent's tape */
    const int base = sh[2];

    for (int i = hi_i; i > lo_i; --i) {
        if (live_of(ch[i]) != 1) continue;
        uint64_t d = tro[root + 1 + i];
        const uint32_t op = (uint32_t)d & 0xFF, r = (uint32_t)(d >> 24) & 0xFF;
        if (mpr_op_is_minmax(op)) {
            const int c = total_choice(ch[i]);
            if (c == 1) d = (d & ~0xFFull) | MPR_OP_COPY_LHS;
            else if (c == 2) d = (d & ~0xFFull) | (r ? MPR_OP_COPY_RHS : MPR_OP_COPY_IMM);
        }
        twr[base + MPR_SUBTAPE_CHUNK * (rank / 62) + 62 - rank % 62] = d;
        ++rank;
    }
    if (tid == 0) {
        twr[base + 63] = tro[root + 1 + n];                       /* end clause */
        /* head: copy of the parent's head, after the last clause written */
        const int last_chunk = nchunks - 1;
        const int in_last = total - 62 * last_chunk;       /* 0 when the last chunk was opened for a dropped clause */
        const int head_at = base + MPR_SUBTAPE_CHUNK * last_chunk + 62 - in_last;
        twr[head_at] = tro[tape];
        a.tiles[gidx].tape = head_at;
        if (a.counters) atomicAdd((unsigned long long*)&a.counters[CNT_WRITTEN], (unsigned long long)(total + 2 + 2 * (nchunks - 1)));
    }
    /* links between consecutive chunks (reference :384-413): word 63 of the newer chunk jumps back
     * to the older one, word 0 of the older chunk forward to the newer */
    for (int k = 1 + tid; k < nchunks; k += nt) {
        const int prev_index = base + MPR_SUBTAPE_CHUNK * (k - 1);
        const int out_index = base + MPR_SUBTAPE_CHUNK * k;
        const int delta = prev_index - (out_index + 63);
        twr[out_index + 63] = (uint64_t)MPR_OP_JUMP | ((uint64_t)(uint32_t)delta << 32);
        twr[prev_index] = (uint64_t)MPR_OP_JUMP | ((uint64_t)(uint32_t)(-delta) << 32);
    }
}

size_t wide_stage_lds_bytes(int nclauses)
{
    return (((size_t)(nclauses + 3) * 8 + (size_t)nclauses + 15) & ~(size_t)15) + 1040 * sizeof(int);
}
bool wide_stage_fits(int nclauses) { return wide_stage_lds_bytes(nclauses) <= 150 * 1024; }
void launch_eval_tiles_wide(hipStream_t s, int dim, const WideStageArgs& w, int threads_forced)
{
    static OncePerDevice once;
    once.run([] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_eval_tiles_wide<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_eval_tiles_wide<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const size_t lds = wide_stage_lds_bytes(w.nclauses);
    /* one wavefront per tile when the levels are narrow (no cross-wave barriers), four otherwise */
    const int width = w.nclauses / (w.nlevels > 0 ? w.nlevels : 1);
    int threads = (width >= 48) ? 256 : 64;
    /* a frame with no more tiles than compute units (256 at 1024^2): sixteen wavefronts per tile shorten
     * both the levels (one clause per thread) and the serial ranking / writing loops */
    if (threads_forced > 0) threads = threads_forced;       /* (development) */
    else if (width >= 192 && w.t.count <= 256) threads = 1024;
    else if (width >= 192 && w.t.count <= 1024) threads = 512;
    /* ... and with thousands of tiles and levels of moderate width two wavefronts per tile (twice the tiles in
     * flight per CU) beat four (architecture, the involute gears at 1024^3 / 4096^2: -10 % on this stage) */
    else if (width >= 48 && width < 192 && w.t.count >= 2048) threads = 128;
    if (dim == 3) hipLaunchKernelGGL(k_eval_tiles_wide<3>, dim3(w.t.count), dim3(threads), lds, s, w);
    else hipLaunchKernelGGL(k_eval_tiles_wide<2>, dim3(w.t.count), dim3(threads), lds, s, w);
}

}  // namespace mprk
