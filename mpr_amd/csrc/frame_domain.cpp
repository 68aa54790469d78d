/* frame_domain.cpp — see frame_domain.hpp */
#include "frame_domain.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>

#include "../../include/mpr_clause.h"

namespace mpr {
namespace {

struct Iv {
    double lo, hi;
};

/* outward to float, and two ulps further (the double routines' own error, and a correctly rounded float result's distance from
 * the double one) */
double down(double d)
{
    float f = (float)d;
    if ((double)f > d) f = std::nextafterf(f, -INFINITY);
    f = std::nextafterf(std::nextafterf(f, -INFINITY), -INFINITY);
    return (double)f;
}
double up(double d)
{
    float f = (float)d;
    if ((double)f < d) f = std::nextafterf(f, INFINITY);
    f = std::nextafterf(std::nextafterf(f, INFINITY), INFINITY);
    return (double)f;
}
Iv out(double lo, double hi) { return {down(lo), up(hi)}; }
bool finite(const Iv& v) { return std::fabs(v.lo) < (double)FLT_MAX && std::fabs(v.hi) < (double)FLT_MAX && v.lo <= v.hi; }

Iv mul(const Iv& a, const Iv& b)
{
    const double p[4] = {a.lo * b.lo, a.lo * b.hi, a.hi * b.lo, a.hi * b.hi};
    return out(*std::min_element(p, p + 4), *std::max_element(p, p + 4));
}
/* (the divisor does not hold zero: checked by the caller) */
Iv div(const Iv& a, const Iv& b)
{
    const double p[4] = {a.lo / b.lo, a.lo / b.hi, a.hi / b.lo, a.hi / b.hi};
    return out(*std::min_element(p, p + 4), *std::max_element(p, p + 4));
}
bool holds_zero(const Iv& v) { return v.lo <= 0.0 && v.hi >= 0.0; }
/* a divisor the device's float routines could see as holding zero: an end below the smallest normal number may round or flush to
 * zero there, and the routine then returns [-inf, inf] (the same margin the logarithm's lower end has) */
bool near_zero(const Iv& v) { return !((v.lo >= (double)FLT_MIN && v.hi >= (double)FLT_MIN) || (v.lo <= -(double)FLT_MIN && v.hi <= -(double)FLT_MIN)); }

}   // namespace

bool frame_is_tame(const uint64_t* cl, int n, int dim, const float* mat, float z, double* trace)
{
    if (!cl || n < 2 || (dim != 2 && dim != 3)) return false;
    if (trace)
        for (int i = 0; i < 2 * n; ++i) trace[i] = NAN;
    /* the view: [-1, 1] on every axis through the matrix, as the tile stages do it (rows of a column-major matrix, then the
     * division by the last row) */
    const Iv unit = {-1.0, 1.0};
    Iv axes[3];
    const int m = dim + 1;
    Iv r[4];
    for (int i = 0; i < m; ++i) {
        double lo = (double)mat[i + m * dim], hi = lo;
        for (int k = 0; k < dim; ++k) {
            const Iv t = mul(unit, Iv{(double)mat[i + m * k], (double)mat[i + m * k]});
            lo += t.lo;
            hi += t.hi;
        }
        r[i] = out(lo, hi);
        if (!finite(r[i])) return false;
    }
    if (near_zero(r[dim])) return false;
    for (int k = 0; k < dim; ++k) {
        axes[k] = div(r[k], r[dim]);
        if (!finite(axes[k])) return false;
    }
    if (dim == 2) axes[2] = Iv{(double)z, (double)z};
    if (!finite(axes[2])) return false;

    Iv slot[256];
    for (Iv& s : slot) s = Iv{0.0, 0.0};
    /* nn[s]: the upper end of slot s is >= 0 in EVERY tile, whatever the tile (a square, a product of a slot with itself, an
     * absolute value, a sum of such ...): sqrt of it never sees an interval that lies below zero, however negative the lower end
     * the interval arithmetic gives x * x over a tile that straddles an axis */
    bool nn[256] = {false};
    slot[mpr_cl_out(cl[0])] = axes[0];
    slot[mpr_cl_lhs(cl[0])] = axes[1];
    slot[mpr_cl_rhs(cl[0])] = axes[2];
    if (trace) {
        trace[0] = axes[0].lo;
        trace[1] = axes[0].hi;
    }
    for (int i = 1; i + 1 < n; ++i) {
        const uint64_t c = cl[i];
        const uint32_t op = mpr_cl_op(c), bits = mpr_cl_immbits(c);
        float immf;
        static_assert(sizeof(immf) == sizeof(bits), "f32");
        __builtin_memcpy(&immf, &bits, 4);
        const Iv K = {(double)immf, (double)immf};
        const Iv a = slot[mpr_cl_lhs(c)], b = slot[mpr_cl_rhs(c)];
        Iv v;
        switch (op) {
            case MPR_OP_SQUARE_LHS: {
                const double p = a.lo * a.lo, q = a.hi * a.hi;
                v = holds_zero(a) ? out(0.0, std::max(p, q)) : out(std::min(p, q), std::max(p, q));
                break;
            }
            case MPR_OP_SQRT_LHS:
                /* (a tile that straddles zero gets [0, sqrt(hi)], inside its parent's; one that lies below gets NaNs) */
                if (a.lo < 0.0 && !nn[mpr_cl_lhs(c)]) return false;
                v = out(std::sqrt(std::max(a.lo, 0.0)), std::sqrt(a.hi));
                if (a.lo <= 0.0) v.lo = 0.0;
                break;
            case MPR_OP_NEG_LHS: v = Iv{-a.hi, -a.lo}; break;
            case MPR_OP_SIN_LHS:
            case MPR_OP_COS_LHS: v = Iv{-1.0, 1.0}; break;                     /* the reference's own (:346-353, :378-380) */
            case MPR_OP_ASIN_LHS:
                if (a.lo < -1.0 || a.hi > 1.0) return false;
                v = out(std::asin(a.lo), std::asin(a.hi));
                break;
            case MPR_OP_ACOS_LHS:
                if (a.lo < -1.0 || a.hi > 1.0) return false;
                v = out(std::acos(a.hi), std::acos(a.lo));
                break;
            case MPR_OP_ATAN_LHS: v = out(std::atan(a.lo), std::atan(a.hi)); break;
            case MPR_OP_EXP_LHS: v = out(std::exp(a.lo), std::exp(a.hi)); break;
            case MPR_OP_ABS_LHS:
                v = holds_zero(a) ? Iv{0.0, std::max(-a.lo, a.hi)} : (a.lo > 0.0 ? a : Iv{-a.hi, -a.lo});
                break;
            case MPR_OP_LOG_LHS:
                /* (a lower end the float routines could round down to zero counts as zero) */
                if (!(a.lo >= (double)FLT_MIN)) return false;
                v = out(std::log(a.lo), std::log(a.hi));
                break;
            case MPR_OP_ADD_LHS_IMM: v = out(a.lo + K.lo, a.hi + K.hi); break;
            case MPR_OP_ADD_LHS_RHS: v = out(a.lo + b.lo, a.hi + b.hi); break;
            case MPR_OP_MUL_LHS_IMM: v = mul(a, K); break;
            case MPR_OP_MUL_LHS_RHS: v = mul(a, b); break;
            case MPR_OP_MIN_LHS_IMM: v = Iv{std::min(a.lo, K.lo), std::min(a.hi, K.hi)}; break;
            case MPR_OP_MIN_LHS_RHS: v = Iv{std::min(a.lo, b.lo), std::min(a.hi, b.hi)}; break;
            case MPR_OP_MAX_LHS_IMM: v = Iv{std::max(a.lo, K.lo), std::max(a.hi, K.hi)}; break;
            case MPR_OP_MAX_LHS_RHS: v = Iv{std::max(a.lo, b.lo), std::max(a.hi, b.hi)}; break;
            case MPR_OP_SUB_LHS_IMM: v = out(a.lo - K.hi, a.hi - K.lo); break;
            case MPR_OP_SUB_IMM_RHS: v = out(K.lo - b.hi, K.hi - b.lo); break;
            case MPR_OP_SUB_LHS_RHS: v = out(a.lo - b.hi, a.hi - b.lo); break;
            case MPR_OP_DIV_LHS_IMM:
                if (near_zero(K)) return false;
                v = div(a, K);
                break;
            case MPR_OP_DIV_IMM_RHS:
                if (near_zero(b)) return false;
                v = div(K, b);
                break;
            case MPR_OP_DIV_LHS_RHS:
                if (near_zero(b)) return false;
                v = div(a, b);
                break;
            case MPR_OP_COPY_IMM: v = K; break;
            case MPR_OP_COPY_LHS: v = a; break;
            case MPR_OP_COPY_RHS: v = b; break;
            default: return false;
        }
        if (!finite(v)) return false;
        const bool na = nn[mpr_cl_lhs(c)], nb = nn[mpr_cl_rhs(c)];
        bool n_out = v.lo >= 0.0;
        switch (op) {
            case MPR_OP_SQUARE_LHS: case MPR_OP_ABS_LHS: case MPR_OP_SQRT_LHS: case MPR_OP_EXP_LHS: case MPR_OP_ACOS_LHS: n_out = true; break;
            case MPR_OP_MUL_LHS_RHS: n_out = n_out || mpr_cl_lhs(c) == mpr_cl_rhs(c); break;
            case MPR_OP_ADD_LHS_RHS: n_out = n_out || (na && nb); break;
            case MPR_OP_ADD_LHS_IMM: n_out = n_out || (na && immf >= 0.0f); break;
            case MPR_OP_MUL_LHS_IMM: n_out = n_out || (na && immf >= 0.0f); break;
            case MPR_OP_MAX_LHS_RHS: n_out = n_out || na || nb; break;
            case MPR_OP_MAX_LHS_IMM: n_out = n_out || na || immf >= 0.0f; break;
            case MPR_OP_MIN_LHS_RHS: n_out = n_out || (na && nb); break;
            case MPR_OP_MIN_LHS_IMM: n_out = n_out || (na && immf >= 0.0f); break;
            case MPR_OP_COPY_LHS: n_out = n_out || na; break;
            case MPR_OP_COPY_RHS: n_out = n_out || nb; break;
            default: break;
        }
        nn[mpr_cl_out(c)] = n_out;
        slot[mpr_cl_out(c)] = v;
        if (trace) {
            trace[2 * i] = v.lo;
            trace[2 * i + 1] = v.hi;
        }
    }
    return true;
}

}   // namespace mpr
