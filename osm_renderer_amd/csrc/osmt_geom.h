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
        /* short edge: (2j+1)a + 2b - 1 < 2^24, both divisions in 32-bit arithmetic */
        const int32_t a32 = (int32_t)a, b32 = (int32_t)b, j32 = (int32_t)j;
        int32_t q0, q1;
        if (a32 >= b32) {
            q0 = (j32 == 0) ? 0 : osmt_ceil_div_pos24((2 * j32 - 1) * a32, 2 * b32);
            q1 = (j32 == b32) ? a32 : osmt_ceil_div_pos24((2 * j32 + 1) * a32, 2 * b32) - 1;
        } else {
            q0 = osmt_udiv24(2 * j32 * a32 + b32, 2 * b32);
            if (j32 == b32) {
                q1 = a32;
            } else {
                q1 = osmt_ceil_div_pos24((2 * j32 + 1) * a32, 2 * b32) - 1;
                if (q1 < q0) q1 = q0;
            }
        }
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
    int64_t sdx, sdy;
    double denom;           /* sqrt(dy^2 + dx^2) */
} osmt_seg;

/* p1 != p2 required (line.rs:73-75 returns early otherwise). */
OSMT_HD void osmt_seg_setup(osmt_seg* s, int32_t p1x, int32_t p1y, int32_t p2x, int32_t p2y, double denom) {
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
    s->sdx = (int64_t)p2x - (int64_t)p1x;
    s->sdy = (int64_t)p2y - (int64_t)p1y;
    s->denom = denom;
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
        const float r2b = osmt_rcp24(2 * b); /* both quotients divide by 2b */
        const int32_t c = osmt_ceil_div_pos24r(2 * a * k - b, 2 * b, r2b);
        const int32_t d = osmt_ceil_div_pos24r(2 * a * c - b, 2 * b, r2b);
        *c_out = c;
        *pe = 2 * a * c - 2 * b * d;
    } else {
        const int64_t c = osmt_corrections(a, b, k);
        const int64_t d = osmt_corrections(a, b, c);
        *c_out = (int32_t)c;
        *pe = (int32_t)(2 * (int64_t)a * c - 2 * (int64_t)b * d);
    }
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
        const float r2a = osmt_rcp24(2 * a); /* both quotients divide by 2a */
        const int32_t c = osmt_udiv24r(2 * b * m - b, 2 * a, r2a) + 1;
        *c_out = c;
        *k_out = osmt_udiv24r(2 * b * c - b, 2 * a, r2a);
        *pe_out = 2 * a * c - 2 * b * m;
    } else {
        const int64_t A = a, B = b, M = m;
        const int64_t c = osmt_udiv(2 * B * M - B, 2 * A) + 1;
        *c_out = (int32_t)c;
        *k_out = (int32_t)osmt_udiv(2 * B * c - B, 2 * A);
        *pe_out = (int32_t)(2 * A * c - 2 * B * M);
    }
}

#endif /* OSMT_GEOM_H */
