/*
 * frame_domain.hpp — is a frame "tame": does every interval operation of the tape stay, in EVERY tile of the frame, on the part of
 * its domain where the reference's interval routines are inclusion-isotone (narrower operands -> a result inside the wider
 * operands' result)?
 *
 * Why it matters.  A frame that starts at the 16^3 tiles with the root tape (context.hip: skip0) instead of walking the 64^3 tiles
 * first and handing their shortened tapes down is the reference's procedure only as long as the interval routines are isotone: a
 * child then decides by itself what its parent would have decided for it — the same, if the child's intervals lie inside the
 * parent's.  What the answer decides (and all it decides): a TAME frame takes that start unverified; a frame that is not takes it
 * too, VERIFIED — the 64^3 tiles are walked beside the frame, every 16^3 tile is held against its parent, the normals pass imposes
 * the parents' decisions, and a frame that fails is rendered again from the 64^3 tiles (kernels.hpp: Skip0ParentsArgs).  (The loose
 * enclosures of frames nobody reads do not ask this question: they are gated by the tape — mpr_tape::loose_ok — and by the loose
 * walk's own request for the exact one, interval_gen.hpp.)
 * The reference's routines (inc/gpu_interval.hpp) are isotone wherever their operands are finite and inside the function's domain,
 * and NOT where a special case takes over: log's lower bound 0 for x.lo <= 0 (:382-390), the NaN ends of asin / acos of an interval
 * that leaves [-1, 1] (:306-324; fmin / fmax then DROP them: a min / max with such an operand returns the other one's bounds), sqrt's
 * NaN below zero, division by an interval that holds zero, anything that overflows.  There the reference's image depends on its
 * hierarchy (what a 64^3 tile decided with a NaN in its bounds binds its children), and only its literal procedure reproduces it
 * (found by tests/test_gpu_fuzz_shapes.py: random shapes with asin / acos leaving their domain inside the view).
 *
 * The test: the tape once, on the host, over the box of the WHOLE view, in double precision with every result widened outward by
 * two float ulps (an enclosure of what the device's correctly rounded float routines return for any tile: tiles' axis intervals lie
 * inside the view's, and on-domain finite interval arithmetic is isotone, by induction over the tape).  A few microseconds per
 * (tape, view); sufficient, not necessary — a shape that leaves a domain somewhere in the view loses the shortcut for the whole frame.
 */
#pragma once
#include <cstdint>

namespace mpr {

/* clauses[0] = the head (axis slots), clauses[n - 1] = the end; mat: column-major 4x4 (dim 3) or 3x3 (dim 2) as the tile stages
 * take it (reference src/context.cu:91-113); z: the 2-D frame's z */
bool frame_is_tame(const uint64_t* clauses, int n, int dim, const float* mat, float z, double* trace = nullptr);
/* trace (tests): 2 * n doubles, the enclosure of every clause's result over the view ([2 i], [2 i + 1]; the head's: the x axis';
 * NaN from the clause that is not tame on) */

}   // namespace mpr
