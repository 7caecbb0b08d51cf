/*
 * osmt_geom.h — closed-form ("jump to row / jump to step") versions of the two
 * integer walks of the reference, shared by the HIP kernels and by a host test
 * shim (tests/ brute-force them against the literal walks of the oracle).
 *
 *   fill:   Zingl-Bresenham edge walk of src/draw/fill.rs:51-104 -> per-row x extent
 *   stroke: Murphy main-axis walk of src/draw/line.rs:85-100,143-157 -> state at step k
 *
 * Coordinates are assumed |c| <= 2^28 (z18 world at scale 2 is 2^27 px wide), so
 * every product below fits in int64.
 */
#ifndef OSMT_GEOM_H
#define OSMT_GEOM_H

#include <stdint.h>

#if defined(__HIPCC__)
#define OSMT_HD __host__ __device__ __forceinline__
#else
#define OSMT_HD inline
#endif

#define OSMT_COORD_LIMIT (1 << 28)

/* floor(n / d) for n >= 0, d > 0.  Uses one f64 division + a remainder fix-up when the
 * numerator is exactly representable (the common case); gfx950 has no integer divider,
 * so this is several times cheaper than the emulated 64-bit division. */
OSMT_HD int64_t osmt_udiv(int64_t n, int64_t d) {
    if (n < ((int64_t)1 << 52)) {
        int64_t q = (int64_t)((double)n / (double)d);
        int64_t r = n - q * d;
        if (r < 0) {
            q -= 1;
        } else if (r >= d) {
            q += 1;
        }
        return q;
    }
    return n / d;
}
/* max(0, ceil(n / d)) for d > 0 */
OSMT_HD int64_t osmt_ceil_div_pos(int64_t n, int64_t d) {
    if (n <= 0) return 0;
    return osmt_udiv(n + d - 1, d);
}

/* floor(n / d) for 0 <= n < 2^24, d > 0 in 32-bit arithmetic: approximate quotient from the
 * f32 reciprocal (v_rcp_f32 on the GPU), then an exact integer remainder fix-up, so the result
 * does not depend on how good the approximation is. */
OSMT_HD float osmt_rcp24(int32_t d) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf((float)d);
#else
    return 1.0f / (float)d;
#endif
}
/* 24-bit multiply (one full-rate instruction on the GPU; both operands and the product fit) */
#if defined(__HIP_DEVICE_COMPILE__)
#define OSMT_MUL24(a, b) __mul24((a), (b))
#else
#define OSMT_MUL24(a, b) ((a) * (b))
#endif
/* floor(n / d) for 0 <= n < 2^24, 0 < d, QUOTIENT < 2^12, reciprocal supplied: with so small a quotient the
 * estimate (float)n * r is within 1e-3 of n/d, so its truncation is off by at most one and ONE correction round
 * is enough (the walks' step / correction counts are all < 2048 on this path). */
OSMT_HD int32_t osmt_udiv24r_small(int32_t n, int32_t d, float r) {
    int32_t q = (int32_t)((float)n * r);
    const int32_t rem = n - OSMT_MUL24(q, d);
    q += (rem >= d) - (rem < 0);
    return q;
}
OSMT_HD int32_t osmt_ceil_div_pos24r_small(int32_t n, int32_t d, float r) {
    if (n <= 0) return 0;
    return osmt_udiv24r_small(n + d - 1, d, r);
}
/* the same with the reciprocal of d supplied (several quotients share one divisor) */
OSMT_HD int32_t osmt_udiv24r(int32_t n, int32_t d, float r) {
    int32_t q = (int32_t)((float)n * r);
    int32_t rem = n - q * d;
    /* |q - floor(n/d)| <= 1 for n < 2^24 (reciprocal and product are each within one f32
     * rounding of the exact value, q <= 2^23); two branch-free correction rounds cover +-2 */
    for (int round = 0; round < 2; ++round) {
        const int32_t up = rem >= d, down = rem < 0;
        q += up - down;
        rem += (down - up) * d;
    }
    return q;
}
OSMT_HD int32_t osmt_udiv24(int32_t n, int32_t d) { return osmt_udiv24r(n, d, osmt_rcp24(d)); }
OSMT_HD int32_t osmt_ceil_div_pos24r(int32_t n, int32_t d, float r) {
    if (n <= 0) return 0;
    return osmt_udiv24r(n + d - 1, d, r);
}
OSMT_HD int32_t osmt_ceil_div_pos24(int32_t n, int32_t d) {
    if (n <= 0) return 0;
    return osmt_udiv24(n + d - 1, d);
}

/* Exact fmod(x, y) for x >= 0, y > 0, x / y < 2^52 (the dash phase `dist_rem %= total_dash_len`,
 * opacity_calculator.rs:57-60).  fmod's result x - n*y (n = trunc(x/y)) is always exactly
 * representable, so one fma with the right n returns it without rounding; the rounded quotient
 * can only be off by one, which the sign / range of the remainder reveals.  Several times
 * shorter than the generic library routine (brute-forced against libm: tests/). */
#if defined(__HIPCC__) || defined(__cplusplus)
#include <math.h>
#endif
OSMT_HD double osmt_fmod_pos(double x, double y) {
    double n = trunc(x / y);
    double r = fma(-n, y, x);
    if (r < 0.0) {
        n -= 1.0;
        r = fma(-n, y, x);
    } else if (r >= y) {
        n += 1.0;
        r = fma(-n, y, x);
    }
    return r;
}

/* Correctly rounded n / d — bit-identical to the IEEE division the reference performs for center_distance
 * (line.rs:116-118) — from the correctly rounded reciprocal r = RN(1 / d), which the pre-pass computes ONCE per
 * segment with a real division: one multiply and two residual corrections, five full-rate instructions instead of
 * the ~12 (one of them the quarter-rate v_rcp_f64) of the generic expansion, per visited pixel.
 * Why it is exact (Markstein; Muller et al., Handbook of Floating-Point Arithmetic, "division with an FMA"): with
 * r within half an ulp of 1/d, q0 = RN(n*r) is within 2 ulps of n/d; e0 = n - d*q0 comes out of the FMA with at
 * most one rounding of a quantity 2^-52 smaller than n, so q1 = RN(q0 + e0*r) is a faithful rounding of n/d; for a
 * faithful q1 the residual e1 = n - d*q1 is exactly representable and q2 = RN(q1 + e1*r) is RN(n/d).  Requires
 * finite operands and no overflow/underflow — here d = |p2 - p1| in [1, 2^30] with n = |cross product| <= 2^60 (any
 * nonzero quotient is >= 2^-30), and d = feather_dist (1.0 up to an ulp) with |n| <= ~2^16.  Every step is odd in n,
 * so a negative numerator gives exactly the negated quotient.  Brute-forced against the hardware division in
 * tests/test_geom_closed_forms.py. */
OSMT_HD double osmt_div_exact(double n, double d, double r) {
    const double q0 = n * r;
    const double e0 = fma(-d, q0, n);
    const double q1 = fma(e0, r, q0);
    const double e1 = fma(-d, q1, n);
    return fma(e1, r, q1);
}

/* ---- fill.rs:51-104 ------------------------------------------------------
 * The walk from p1 to p2 visits, on the row reached after j y-steps (0 <= j <= DY),
 * the columns i_first(j) .. i_last(j) (counted in x-steps from p1).  With a = |dx|,
 * b = DY = |dy| > 0 and L(j) = ceil((2j+1)a / 2b) - 1:
 *   x-major (a >= b): i_first(0) = 0, i_first(j) = L(j-1) + 1;  i_last(j) = L(j), i_last(b) = a
 *   y-major (a <  b): i_first(j) = floor((2ja + b) / 2b);       i_last(j) = max(i_first, L(j)), i_last(b) = a
 * (brute-forced against the literal walk: tests/test_geom_closed_forms.py).
 *
 * Returns 0 when the edge has no un-poisoned record on row y (row outside the edge, the
 * row of the smaller-y endpoint — fill.rs:66-72 — or a horizontal edge), else 1 and the
 * Edge{x_min, x_max} of fill.rs:79-87. */
OSMT_HD int osmt_fill_row_extent(int32_t p1x, int32_t p1y, int32_t p2x, int32_t p2y, int32_t y, int32_t* x_min,
                                 int32_t* x_max) {
    const int32_t ytop = p1y < p2y ? p1y : p2y;
    const int32_t ybot = p1y < p2y ? p2y : p1y;
    if (y <= ytop || y > ybot) return 0;
    const int64_t a = p2x >= p1x ? (int64_t)p2x - p1x : (int64_t)p1x - p2x;
    const int64_t b = (int64_t)ybot - ytop;
    const int32_t sx = p1x < p2x ? 1 : -1;
    const int64_t j = p1y < p2y ? (int64_t)y - p1y : (int64_t)p1y - y;
    int64_t i0, i1;
    if (a < 2048 && b < 2048) {
        /* short edge: (2j+1)a + 2b - 1 < 2^24 and every quotient is <= a < 2^11: both divisions share the f32
         * reciprocal of 2b and need one correction round (osmt_udiv24r_small).  Branch-free in the edge's slope — the
         * lanes of a wave hold different edges, and a branch on x-major / y-major would run both sides for all:
         *   first = floor(N0 / 2b),  N0 = x-major ? (2j-1)a + 2b - 1  (= ceil((2j-1)a / 2b), 0 for j = 0)
         *                                         : 2ja + b
         *   last  = j == b ? a : ceil((2j+1)a / 2b) - 1, and not below `first` for a y-major edge. */
        const int32_t a32 = (int32_t)a, b32 = (int32_t)b, j32 = (int32_t)j;
        const float r2b = osmt_rcp24(2 * b32);
        const int32_t xmajor = a32 >= b32;
        const int32_t ja2 = OSMT_MUL24(2 * j32, a32);
        int32_t n0 = xmajor ? ja2 - a32 + 2 * b32 - 1 : ja2 + b32;
        if (xmajor && ja2 - a32 <= 0) n0 = 0; /* osmt_ceil_div_pos: a non-positive numerator gives 0 (j = 0, or a = 0) */
        const int32_t q0 = osmt_udiv24r_small(n0, 2 * b32, r2b);
        const int32_t n1 = ja2 + a32; /* (2j+1)a >= 0 */
        int32_t q1 = (n1 <= 0 ? 0 : osmt_udiv24r_small(n1 + 2 * b32 - 1, 2 * b32, r2b)) - 1;
        if (!xmajor && q1 < q0) q1 = q0;
        if (j32 == b32) q1 = a32;
        i0 = q0;
        i1 = q1;
    } else if (a >= b) {

        i0 = (j == 0) ? 0 : osmt_ceil_div_pos((2 * j - 1) * a, 2 * b); /* L(j-1) + 1 */
        i1 = (j == b) ? a : osmt_ceil_div_pos((2 * j + 1) * a, 2 * b) - 1;
    } else {
        i0 = osmt_udiv(2 * j * a + b, 2 * b);
        if (j == b) {
            i1 = a;
        } else {
            i1 = osmt_ceil_div_pos((2 * j + 1) * a, 2 * b) - 1;
            if (i1 < i0) i1 = i0;
        }
    }
    const int32_t xa = p1x + sx * (int32_t)i0;
    const int32_t xb = p1x + sx * (int32_t)i1;
    *x_min = xa < xb ? xa : xb;
    *x_max = xa < xb ? xb : xa;
    return 1;
}

/* ---- line.rs:65-158 --------------------------------------------------------
 * Per-segment constants (line.rs:75-104). */
typedef struct osmt_seg {
    int32_t p1x, p1y, p2x, p2y;
    int32_t swap;           /* dx > dy: x is the major axis */
    int32_t mn0, mx0;       /* minor / major coordinate of p1 */
    int32_t a, b;           /* mn_delta, mx_delta  (0 <= a <= b, b >= 1) */
    int32_t mn_inc, mx_inc; /* get_inc: from <= to ? 1 : -1 */
    int64_t numer_const;    /* p2.x*p1.y - p2.y*p1.x */
    int32_t sdx, sdy;       /* p2 - p1 (|.| <= 2^29: coordinates are limited to 2^28) */
    double denom;           /* sqrt(dy^2 + dx^2) */
    double rdenom;          /* RN(1 / denom) */
} osmt_seg;

/* p1 != p2 required (line.rs:73-75 returns early otherwise). */
OSMT_HD void osmt_seg_setup(osmt_seg* s, int32_t p1x, int32_t p1y, int32_t p2x, int32_t p2y, double denom, double rdenom) {
    const int32_t dx = p2x >= p1x ? p2x - p1x : p1x - p2x;
    const int32_t dy = p2y >= p1y ? p2y - p1y : p1y - p2y;
    const int32_t incx = p1x <= p2x ? 1 : -1;
    const int32_t incy = p1y <= p2y ? 1 : -1;
    s->p1x = p1x;
    s->p1y = p1y;
    s->p2x = p2x;
    s->p2y = p2y;
    s->swap = dx > dy;
    s->mn0 = s->swap ? p1y : p1x;
    s->mx0 = s->swap ? p1x : p1y;
    s->a = s->swap ? dy : dx;
    s->b = s->swap ? dx : dy;
    s->mn_inc = s->swap ? incy : incx;
    s->mx_inc = s->swap ? incx : incy;
    s->numer_const = (int64_t)p2x * (int64_t)p1y - (int64_t)p2y * (int64_t)p1x;
    s->sdx = p2x - p1x;
    s->sdy = p2y - p1y;
    s->denom = denom;
    s->rdenom = rdenom;
}

/* osmt_stroke_step for short segments (b < 2048): every intermediate is < 2^23, so the two
 * ceil-divisions run in 32-bit arithmetic (gfx950 has no integer divider; the generic path
 * below goes through f64).  Same outputs as osmt_stroke_step. */
#define OSMT_STEP24_MAX_B 2048
OSMT_HD void osmt_stroke_step24(int32_t a, int32_t b, int32_t k, int32_t* c_out, int32_t* pe, int32_t* has_extra,
                                int32_t* pe_extra) {
    const int32_t c = osmt_ceil_div_pos24(2 * a * k - b, 2 * b);
    const int32_t d = osmt_ceil_div_pos24(2 * a * c - b, 2 * b);
    const int32_t pe_now = 2 * a * c - 2 * b * d;
    *c_out = c;
    *pe = pe_now;
    *has_extra = 0;
    *pe_extra = 0;
    if (k < b) {
        const int32_t e = 2 * a * k - 2 * b * c;
        if (e + 2 * a > b && pe_now + 2 * a > b) {
            *has_extra = 1;
            *pe_extra = pe_now - 2 * b + 2 * a;
        }
    }
}

/* Number of corrections after k calls of update_error (line.rs:91-100) starting from 0:
 * c_k = max(0, ceil((2ak - b) / 2b)). */
OSMT_HD int64_t osmt_corrections(int64_t a, int64_t b, int64_t k) { return osmt_ceil_div_pos(2 * a * k - b, 2 * b); }

/* State of the main loop (line.rs:143-157) at main-axis step k (0 <= k <= b):
 *   main perpendiculars at (mn, mx) = (mn0 + c*mn_inc, mx0 + k*mx_inc) with p_error = pe;
 *   when *has_extra, a second pair of perpendiculars (line.rs:152-154) at
 *   (mn + mn_inc, mx) with p_error = *pe_extra. */
OSMT_HD void osmt_stroke_step(int32_t a32, int32_t b32, int32_t k32, int32_t* c_out, int32_t* pe, int32_t* has_extra,
                              int32_t* pe_extra) {
    const int64_t a = a32, b = b32, k = k32;
    const int64_t c = osmt_corrections(a, b, k);
    const int64_t d = osmt_corrections(a, b, c);
    *c_out = (int32_t)c;
    *pe = (int32_t)(2 * a * c - 2 * b * d);
    *has_extra = 0;
    *pe_extra = 0;
    if (k < b) {
        /* update_error(error) corrects at this step iff e_k + 2a > b, e_k = 2ak - 2bc */
        const int64_t e = 2 * a * k - 2 * b * c;
        if (e + 2 * a > b) {
            const int64_t pe_now = 2 * a * c - 2 * b * d;
            if (pe_now + 2 * a > b) { /* update_error(p_error) corrects too */
                *has_extra = 1;
                *pe_extra = (int32_t)(pe_now - 2 * b + 2 * a);
            }
        }
    }
}

/* Main perpendicular of step k only: c = corrections so far, pe = p_error (see osmt_stroke_step). */
OSMT_HD void osmt_stroke_main(int32_t a, int32_t b, int32_t k, int32_t* c_out, int32_t* pe) {
    if (b < OSMT_STEP24_MAX_B) {
        const float r2b = osmt_rcp24(2 * b); /* both quotients divide by 2b; a, k, c, d <= b < 2048 */
        const int32_t c = osmt_ceil_div_pos24r_small(OSMT_MUL24(2 * a, k) - b, 2 * b, r2b);
        const int32_t d = osmt_ceil_div_pos24r_small(OSMT_MUL24(2 * a, c) - b, 2 * b, r2b);
        *c_out = c;
        *pe = OSMT_MUL24(2 * a, c) - OSMT_MUL24(2 * b, d);
    } else {
        const int64_t c = osmt_corrections(a, b, k);
        const int64_t d = osmt_corrections(a, b, c);
        *c_out = (int32_t)c;
        *pe = (int32_t)(2 * (int64_t)a * c - 2 * (int64_t)b * d);
    }
}

/* ---- how far a perpendicular run can draw (line.rs:108-137) -------------------------------------------------
 * A run starts on a Bresenham centre (main run) or one minor step beside it (extra run, line.rs:152-154) and moves
 * one pixel per step t along the segment's MINOR axis, with cc_t <= t*a/b + 1 corrections of one pixel along the
 * major axis.  The pixel of step t lies at distance >= t*len/b - a/len - |d0| from the ideal line (both kinds of
 * move increase the distance; the corrections lag the ideal perpendicular by less than one, which is the a/len),
 * where |d0| <= 0.5 for a main run and <= 1.5 for an extra one, and it is set only while that distance is below
 * feather_to <= ft (opacity_calculator.rs:171-185).  Hence
 *     t < (ft + a/len + |d0|) * b/len,        cc_t <= t_max * a/b + 1.
 * ft must be max(|half_width| + 0.5, 1.0): the calculator works with sqrt(h*h - cap_dist^2), i.e. with |h|. */
typedef struct osmt_run_reach {
    int32_t t_main, c_main;   /* main runs: steps 0 .. t_main, major offsets 0 .. c_main */
    int32_t t_extra, c_extra; /* extra runs */
} osmt_run_reach;

#define OSMT_REACH_MAX 60000 /* widths are validated so that no bound comes near it */
OSMT_HD osmt_run_reach osmt_reach_of(int32_t a, int32_t b, double len, double ft) {
    osmt_run_reach r;
    const double al = (double)a / len, bl = (double)b / len;
    /* floor(x) >= the largest integer t with t < x; the 1e-9 absorbs the roundings of the few operations above */
    const double xm = (ft + al + 0.5) * bl + 1e-9, xe = (ft + al + 1.5) * bl + 1e-9;
    const double tm = floor(xm), te = floor(xe);
    const double ab = (double)a / (double)b;
    const double cm = floor(tm * ab + 1e-9) + 1.0, ce = floor(te * ab + 1e-9) + 1.0;
    r.t_main = (int32_t)(tm < (double)OSMT_REACH_MAX ? tm : (double)OSMT_REACH_MAX);
    r.t_extra = (int32_t)(te < (double)OSMT_REACH_MAX ? te : (double)OSMT_REACH_MAX);
    r.c_main = (int32_t)(cm < (double)OSMT_REACH_MAX ? cm : (double)OSMT_REACH_MAX);
    r.c_extra = (int32_t)(ce < (double)OSMT_REACH_MAX ? ce : (double)OSMT_REACH_MAX);
    return r;
}

/* Smallest main-axis step k (0 <= k) whose correction count c_k = osmt_corrections(a, b, k) is >= c:
 * c_k >= c  <=>  2ak > 2bc - b  <=>  k >= floor((2bc - b) / 2a) + 1   (c >= 1, a >= 1). */
OSMT_HD int64_t osmt_first_step_with_corrections(int32_t a, int32_t b, int64_t c) {
    if (c <= 0) return 0;
    return osmt_udiv(2 * (int64_t)b * c - b, 2 * (int64_t)a) + 1;
}

/* ---- the extra perpendiculars of line.rs:152-154 as directly enumerable events -------------
 * An extra pair fires at step k < b exactly when both the main error and p_error are corrected
 * (c and d = corrections(c) both increment).  d grows by at most one per correction, hence:
 *   - the number of events at steps k < K is E(K) = d(c_K);
 *   - the m-th event (m >= 1, needs a > 0) happens at c = c_m = floor((2bm - b) / 2a) + 1 (the
 *     smallest c with d(c) >= m), on step k = floor((2b*c_m - b) / 2a) (the step whose update
 *     brings the correction count to c_m), at (mn0 + c_m*mn_inc, mx0 + k*mx_inc), with
 *     p_error = 2a*c_m - 2b*m.
 * Brute-forced against the literal loop in tests/test_geom_closed_forms.py. */
OSMT_HD int32_t osmt_extra_count(int32_t a, int32_t b, int32_t K) {
    if (a <= 0 || K <= 0) return 0;
    if (b < OSMT_STEP24_MAX_B) {
        const int32_t c = osmt_ceil_div_pos24(2 * a * K - b, 2 * b);
        return osmt_ceil_div_pos24(2 * a * c - b, 2 * b);
    }
    const int64_t c = osmt_corrections(a, b, K);
    return (int32_t)osmt_corrections(a, b, c);
}
OSMT_HD void osmt_extra_event(int32_t a, int32_t b, int32_t m, int32_t* c_out, int32_t* k_out, int32_t* pe_out) {
    if (b < OSMT_STEP24_MAX_B) {
        const float r2a = osmt_rcp24(2 * a); /* both quotients divide by 2a; m <= a, c, k <= b < 2048 */
        const int32_t c = osmt_udiv24r_small(OSMT_MUL24(2 * b, m) - b, 2 * a, r2a) + 1;
        *c_out = c;
        *k_out = osmt_udiv24r_small(OSMT_MUL24(2 * b, c) - b, 2 * a, r2a);
        *pe_out = OSMT_MUL24(2 * a, c) - OSMT_MUL24(2 * b, m);
    } else {
        const int64_t A = a, B = b, M = m;
        const int64_t c = osmt_udiv(2 * B * M - B, 2 * A) + 1;
        *c_out = (int32_t)c;
        *k_out = (int32_t)osmt_udiv(2 * B * c - B, 2 * A);
        *pe_out = (int32_t)(2 * A * c - 2 * B * M);
    }
}

/* osmt_corrections / osmt_first_step_with_corrections with the 32-bit fast path of short segments (b < 2048 and the
 * count argument <= b + 2: every numerator stays below 2^24) */
OSMT_HD int32_t osmt_corrections_any(int32_t a, int32_t b, int32_t k) {
    if (b < OSMT_STEP24_MAX_B) return osmt_ceil_div_pos24(2 * a * k - b, 2 * b);
    return (int32_t)osmt_corrections(a, b, k);
}
OSMT_HD int64_t osmt_first_step_any(int32_t a, int32_t b, int32_t c) {
    if (c <= 0) return 0;
    if (b < OSMT_STEP24_MAX_B) return (int64_t)osmt_udiv24(2 * b * c - b, 2 * a) + 1;
    return osmt_first_step_with_corrections(a, b, c);
}

/* the item ranges of one (segment, sub-tile) pair, see osmt_seg_ranges */
typedef struct osmt_item_ranges {
    int32_t k_lo0, k_n0, k_lo1, k_n1; /* main perpendiculars: steps [k_lo, k_lo + k_n) per side */
    int32_t m_lo0, n_x0, m_lo1, n_x1; /* extra perpendiculars (line.rs:152-154): events [m_lo, m_lo + n_x) per side */
} osmt_item_ranges;
#define OSMT_MAX(a, b) ((a) > (b) ? (a) : (b))
#define OSMT_MIN(a, b) ((a) < (b) ? (a) : (b))

/* Items of segment p1->p2 for this sub-tile: per side the main-axis steps [k_lo, k_lo + k_n) whose perpendicular
 * run can reach the sub-tile, and the extra-perpendicular events (line.rs:152-154) that can; returns the total item
 * count (0 when culled).  Every item is exactly ONE perpendicular run.
 * The run of step k on side `mul` starts at (mn_k, mx_k) = (mn0 + c_k*mn_inc, mx0 + k*mx_inc), c_k = corrections
 * after k steps, and its pixel of step t is (mn_k + mul*mn_inc*t, mx_k - mul*mx_inc*cc), 0 <= t <= T, 0 <= cc <= C
 * (osmt_reach_of).  Both mx_k and mn_k are monotonic in k, so "the start lies within reach of the rectangle" is a
 * range of k along the major axis AND a range of c_k — i.e. again a range of k — along the minor axis; the items
 * are the intersection.  Extra event m sits at (c_m, k_m), both monotonic in m, and is clipped the same way. */
OSMT_HD uint32_t osmt_seg_ranges(int32_t p1x, int32_t p1y, int32_t p2x, int32_t p2y, double len, double ft, int32_t rx0,
                                 int32_t ry0, int32_t rx1, int32_t ry1, osmt_item_ranges* q) {
    q->k_lo0 = q->k_n0 = q->k_lo1 = q->k_n1 = 0;
    q->m_lo0 = q->n_x0 = q->m_lo1 = q->n_x1 = 0;
    if (p1x == p2x && p1y == p2y) return 0u; /* line.rs:73-75 */
    const int32_t dx = p2x >= p1x ? p2x - p1x : p1x - p2x, dy = p2y >= p1y ? p2y - p1y : p1y - p2y;
    const bool swap = dx > dy;
    const int32_t bmax = swap ? dx : dy, amin = swap ? dy : dx;
    const osmt_run_reach rr = osmt_reach_of(amin, bmax, len, ft);
    {
        /* every set pixel lies within (t_extra, c_extra) of the segment's box along its (minor, major) axis */
        const int32_t gx = swap ? rr.c_extra : rr.t_extra, gy = swap ? rr.t_extra : rr.c_extra;
        if (OSMT_MAX(p1x, p2x) + gx < rx0 || OSMT_MIN(p1x, p2x) - gx > rx1 || OSMT_MAX(p1y, p2y) + gy < ry0 || OSMT_MIN(p1y, p2y) - gy > ry1)
            return 0u;
    }
    const int32_t mx0 = swap ? p1x : p1y, mn0 = swap ? p1y : p1x;
    const int32_t mx_inc = swap ? (p1x <= p2x ? 1 : -1) : (p1y <= p2y ? 1 : -1);
    const int32_t mn_inc = swap ? (p1y <= p2y ? 1 : -1) : (p1x <= p2x ? 1 : -1);
    const int32_t LO = swap ? rx0 : ry0, HI = swap ? rx1 : ry1;     /* rectangle along the major axis */
    const int32_t mLO = swap ? ry0 : rx0, mHI = swap ? ry1 : rx1;   /* ... along the minor axis */
    const int32_t c_end = osmt_corrections_any(amin, bmax, bmax);       /* c_k <= c_end */
    uint32_t total = 0;
    for (int side = 0; side < 2; ++side) {
        const int32_t mul = side ? -1 : 1;
        int32_t k_lo = 0, k_n = 0, m_lo = 0, n_x = 0;
        for (int extra = 0; extra < 2; ++extra) {
            const int32_t T = extra ? rr.t_extra : rr.t_main, C = extra ? rr.c_extra : rr.c_main;
            /* major axis: pixel major = mx_k - mul*mx_inc*cc */
            int32_t lo = LO, hi = HI;
            if (mul * mx_inc > 0) hi += C; else lo -= C;
            int32_t ka = (mx_inc > 0) ? lo - mx0 : mx0 - hi;
            int32_t kb = (mx_inc > 0) ? hi - mx0 : mx0 - lo;
            ka = OSMT_MAX(ka, 0);
            kb = OSMT_MIN(kb, bmax);
            /* minor axis: pixel minor = mn_k + mul*mn_inc*t  ->  start count c in [cl, ch] */
            int32_t l2 = mLO, h2 = mHI;
            if (mul * mn_inc > 0) l2 -= T; else h2 += T;
            int32_t cl = (mn_inc > 0) ? l2 - mn0 : mn0 - h2;
            int32_t ch = (mn_inc > 0) ? h2 - mn0 : mn0 - l2;
            cl = OSMT_MAX(cl, 0);
            ch = OSMT_MIN(ch, c_end + 1); /* an extra run starts one correction beyond its step's count */
            if (ka > kb || cl > ch) continue;
            if (!extra) {
                /* steps with c_k in [cl, ch]: first k with c_k >= cl .. last k with c_k <= ch */
                int64_t k1 = ka, k2 = kb;
                if (amin > 0) {
                    k1 = OSMT_MAX((int64_t)ka, osmt_first_step_any(amin, bmax, cl));
                    k2 = OSMT_MIN((int64_t)kb, osmt_first_step_any(amin, bmax, ch + 1) - 1);
                } else if (cl > 0) {
                    continue; /* axis-parallel segment: c_k == 0 on every step */
                }
                if (k1 > k2) continue;
                k_lo = (int32_t)k1;
                k_n = (int32_t)(k2 - k1 + 1);
            } else {
                if (amin <= 0) continue; /* no extra perpendiculars without corrections */
                /* events on steps ka .. OSMT_MIN(kb, bmax-1): E(OSMT_MIN(kb, bmax-1) + 1) - E(ka), numbered from 1 */
                const int32_t e0 = osmt_extra_count(amin, bmax, ka);
                const int32_t e1 = osmt_extra_count(amin, bmax, OSMT_MIN(kb, bmax - 1) + 1);
                /* events with start count c_m in [cl, ch]: #events with c_m <= c is d(c) = corrections(a, b, c) */
                const int32_t f0 = (cl > 0) ? osmt_corrections_any(amin, bmax, cl - 1) : 0;
                const int32_t f1 = osmt_corrections_any(amin, bmax, ch);
                const int32_t ma = OSMT_MAX(e0, f0) + 1, mb = OSMT_MIN(e1, f1);
                if (ma > mb) continue;
                m_lo = ma;
                n_x = mb - ma + 1;
            }
        }
        if (side == 0) {
            q->k_lo0 = k_lo; q->k_n0 = k_n; q->m_lo0 = m_lo; q->n_x0 = n_x;
        } else {
            q->k_lo1 = k_lo; q->k_n1 = k_n; q->m_lo1 = m_lo; q->n_x1 = n_x;
        }
        total += (uint32_t)(k_n + n_x);
    }
    return total;
}

#endif /* OSMT_GEOM_H */
